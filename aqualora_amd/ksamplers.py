"""Sigma-space ("k-diffusion") samplers of evaluation/utils_eval.py:83-101 on the HIP U-Net: ``euler`` (EulerDiscreteScheduler),
``heun`` (HeunDiscreteScheduler), ``kdpm2`` (KDPM2DiscreteScheduler) and ``lms`` (LMSDiscreteScheduler), restated from the
published algorithms (Karras et al. 2022, Alg. 1 / k-diffusion ``sample_euler``, ``sample_heun``, ``sample_dpm_2``, ``sample_lms``) with
the conventions diffusers applies to the SD-1.5 scheduler config: ``timestep_spacing="leading"``, ``steps_offset=1``, epsilon
prediction, sigma_t = sqrt((1 - acp_t) / acp_t), a final sigma of 0, ``init_noise_sigma = sqrt(sigma_max^2 + 1)``, and
``scale_model_input``: the U-Net sees x / sqrt(sigma^2 + 1).  diffusers is not on disk: UNPINNED (the property tests in
tests/test_samplers.py hold for any consistent solver; the step coefficients are checked against an independent restatement in
oracle/ppft_oracle.py).  Round 3, third session: ``kdpm2a`` (KDPM2AncestralDiscreteScheduler = k-diffusion ``sample_dpm_2_ancestral``, eta = 1,
noise from a caller-supplied function) and ``pndm`` (PNDMScheduler with the SD-1.5 config's ``skip_prk_steps=True``: the PLMS
linear-multistep method of Liu et al. 2022 with diffusers' warm-up -- the second timestep is visited twice -- and its
``_get_prev_sample`` transfer formula), both UNPINNED like the rest.  Still not built: dpms_s, dpms_sde (needs torchsde's Brownian
tree to reproduce the reference's noise), unipc.

Host logic in float64, the state x in fp32 on the device; the update between two U-Net calls is a handful of element-wise
torch ops on a [B,4,64,64] tensor (plumbing next to a 5 ms U-Net call).  DDIM and DPM-Solver++(2M) live in inference.py.
"""
import math

import torch

from .inference import _cfg_scale, ddim_timesteps
from .watermark import sd15_alphas_cumprod

SAMPLERS = ("euler", "heun", "kdpm2", "lms", "kdpm2a")


def k_sigma_table():
    """sigma of every training timestep 0..999, float64."""
    acp = sd15_alphas_cumprod(device="cpu").double()
    return ((1 - acp) / acp).sqrt()


def k_schedule(num_inference_steps):
    """(timesteps [N] descending ints, sigmas [N + 1] float64 with the trailing 0) for the SD-1.5 scheduler config."""
    ts = ddim_timesteps(num_inference_steps)
    table = k_sigma_table()
    sig = torch.cat([table[torch.tensor(ts)], torch.zeros(1, dtype=torch.float64)])
    return ts, sig


def sigma_to_t(sigma, table=None):
    """Fractional training timestep of a sigma: linear interpolation in log-sigma over the table (k-diffusion ``sigma_to_t``,
    used by the midpoint of KDPM2)."""
    table = k_sigma_table() if table is None else table
    ls = table.log()
    x = math.log(max(float(sigma), 1e-10))
    hi = int(torch.searchsorted(ls, torch.tensor(x, dtype=torch.float64)).clamp(1, len(ls) - 1))
    lo = hi - 1
    w = (x - float(ls[lo])) / (float(ls[hi]) - float(ls[lo]))
    return min(max(lo + w, 0.0), len(ls) - 1.0)


def lms_coefficient(sigmas, order, i, j, n=2001):
    """Integral over [sigma_i, sigma_{i+1}] of the Lagrange basis polynomial of node j among the `order` most recent sigmas
    (k-diffusion ``linear_multistep_coeff``); composite Simpson on n points (the integrand is a polynomial of degree < order)."""
    a, b = float(sigmas[i]), float(sigmas[i + 1])
    tau = torch.linspace(a, b, n, dtype=torch.float64)
    prod = torch.ones_like(tau)
    for k in range(order):
        if k == j:
            continue
        prod = prod * (tau - float(sigmas[i - k])) / (float(sigmas[i - j]) - float(sigmas[i - k]))
    h = (b - a) / (n - 1)
    w = torch.ones(n, dtype=torch.float64)
    w[1:-1:2], w[2:-1:2] = 4.0, 2.0
    return float((prod * w).sum() * h / 3.0)


def ancestral_step(sigma, sigma_next, eta=1.0):
    """(sigma_down, sigma_up) of an ancestral step (k-diffusion ``get_ancestral_step``): integrate down to sigma_down, then add
    sigma_up of fresh noise so that the marginal noise level is sigma_next."""
    if sigma_next <= 0:
        return 0.0, 0.0
    up = min(sigma_next, eta * math.sqrt(sigma_next ** 2 * (sigma ** 2 - sigma_next ** 2) / sigma ** 2))
    return math.sqrt(sigma_next ** 2 - up ** 2), up


def k_sample_core(eps_fn, x, timesteps, sigmas, sampler="euler", lms_order=4, noise_fn=None):
    """Integrate dx/dsigma = eps(x, sigma) from sigmas[0] to 0.  ``eps_fn(x, sigma, t)`` returns the (guided) noise prediction for
    the UN-scaled state x at noise level sigma / fractional timestep t.  x: the state in k-space (x0 + sigma * noise).
    ``noise_fn(i, x)`` supplies the fresh N(0, 1) noise of ancestral step i (kdpm2a)."""
    if sampler not in SAMPLERS:
        raise ValueError(f"sampler {sampler!r} is not one of {SAMPLERS}")
    if sampler == "kdpm2a" and noise_fn is None:
        raise ValueError("kdpm2a is an ancestral sampler: pass noise_fn(i, x)")
    table = k_sigma_table() if sampler in ("kdpm2", "kdpm2a") else None
    hist = []
    n = len(timesteps)
    for i in range(n):
        s, sn, t = float(sigmas[i]), float(sigmas[i + 1]), float(timesteps[i])
        d = eps_fn(x, s, t)
        if sampler == "euler":
            x = x + d * (sn - s)
        elif sampler == "heun":
            x2 = x + d * (sn - s)
            if sn > 0:
                d2 = eps_fn(x2, sn, float(timesteps[i + 1]))
                x = x + (d + d2) * (0.5 * (sn - s))
            else:
                x = x2
        elif sampler == "kdpm2":
            if sn > 0:
                sm = math.exp(0.5 * (math.log(s) + math.log(sn)))
                xm = x + d * (sm - s)
                d2 = eps_fn(xm, sm, sigma_to_t(sm, table))
                x = x + d2 * (sn - s)
            else:
                x = x + d * (sn - s)
        elif sampler == "kdpm2a":
            down, up = ancestral_step(s, sn)
            if down == 0:
                x = x + d * (down - s)
            else:
                sm = math.exp(0.5 * (math.log(s) + math.log(down)))
                xm = x + d * (sm - s)
                d2 = eps_fn(xm, sm, sigma_to_t(sm, table))
                x = x + d2 * (down - s)
                x = x + noise_fn(i, x) * up
        else:   # lms
            hist.append(d)
            if len(hist) > lms_order:
                hist.pop(0)
            order = len(hist)
            upd = None
            for j in range(order):
                c = lms_coefficient(sigmas, order, i, j)
                term = hist[-1 - j] * c
                upd = term if upd is None else upd + term
            x = x + upd
    return x


# ------------------------------------------------------------------------------------------------ PNDM (PLMS)
def pndm_timesteps(num_inference_steps, num_train_timesteps=1000, steps_offset=1):
    """The PLMS timestep list of PNDMScheduler.set_timesteps with skip_prk_steps: the "leading" grid, its second entry visited twice
    (N + 1 model evaluations for N steps)."""
    ratio = num_train_timesteps // num_inference_steps
    base = [i * ratio + steps_offset for i in range(num_inference_steps)]          # ascending
    return (base[:-1] + base[-2:-1] + base[-1:])[::-1], ratio


def pndm_prev_sample(x, t, t_prev, eps, acp):
    """PNDMScheduler._get_prev_sample (formula (9) of the PNDM paper), epsilon prediction, final alpha = alphas_cumprod[0]."""
    a_t = float(acp[t])
    a_p = float(acp[t_prev]) if t_prev >= 0 else float(acp[0])
    coeff = math.sqrt(a_p / a_t)
    denom = a_t * math.sqrt(1 - a_p) + math.sqrt(a_t * (1 - a_t) * a_p)
    return x * coeff - eps * ((a_p - a_t) / denom)


def pndm_sample_core(eps_fn, x, num_inference_steps, acp=None):
    """PLMS loop (PNDMScheduler.step_plms).  ``eps_fn(x, t)`` -> (guided) noise prediction at integer timestep t."""
    acp = sd15_alphas_cumprod(device="cpu").double() if acp is None else acp
    ts, ratio = pndm_timesteps(num_inference_steps)
    ets, cur = [], None
    for counter, t in enumerate(ts):
        e = eps_fn(x, t)
        t_prev = t - ratio
        if counter != 1:
            ets = ets[-3:] + [e]
        else:                      # the repeated timestep: redo the first step with the averaged slope
            t_prev, t = t, t + ratio
        if len(ets) == 1 and counter == 0:
            cur = x
        elif len(ets) == 1 and counter == 1:
            e = (e + ets[-1]) / 2
            x, cur = cur, None
        elif len(ets) == 2:
            e = (3 * ets[-1] - ets[-2]) / 2
        elif len(ets) == 3:
            e = (23 * ets[-1] - 16 * ets[-2] + 5 * ets[-3]) / 12
        else:
            e = (55 * ets[-1] - 59 * ets[-2] + 37 * ets[-3] - 9 * ets[-4]) / 24
        x = pndm_prev_sample(x, t, t_prev, e, acp)
    return x


@torch.no_grad()
def pndm_sample(unet, ctx_cond, ctx_uncond, latents, num_inference_steps=50, guidance_scale=7.5, scale=None):
    """``pndm`` of evaluation/utils_eval.py:91-92 on the HIP U-Net: latents [B,4,h,w] fp32 ~ N(0,1) (init_noise_sigma = 1, no input
    scaling) -> final fp32 latents."""
    dev = latents.device
    B = latents.shape[0]
    ctx = torch.cat([ctx_uncond, ctx_cond]).to(torch.bfloat16).contiguous()
    scale2 = _cfg_scale(scale, B)

    def eps_fn(x, t):
        tt = torch.full((2 * B,), int(t), dtype=torch.long, device=dev)
        xin = x.contiguous()
        e = unet(torch.cat([xin, xin]), tt, ctx, cross_attention_kwargs={"scale": scale2}).sample.float()
        return e[:B] + guidance_scale * (e[B:] - e[:B])

    return pndm_sample_core(eps_fn, latents.float(), num_inference_steps)


@torch.no_grad()
def k_sample(unet, ctx_cond, ctx_uncond, latents, sampler="euler", num_inference_steps=50, guidance_scale=7.5, scale=None, generator=None):
    """latents [B,4,h,w] fp32 ~ N(0,1) -> final fp32 latents (x0 estimate at sigma = 0).  One guided U-Net call (batch 2B) per model
    evaluation: ``num_inference_steps`` of them for euler / lms, twice that minus one for heun / kdpm2."""
    ts, sig = k_schedule(num_inference_steps)
    dev = latents.device
    B = latents.shape[0]
    ctx = torch.cat([ctx_uncond, ctx_cond]).to(torch.bfloat16).contiguous()
    scale2 = _cfg_scale(scale, B)

    def eps_fn(x, sigma, t):
        inp = (x / math.sqrt(sigma * sigma + 1.0)).contiguous()
        tt = torch.full((2 * B,), float(t), dtype=torch.float32 if float(t) != int(t) else torch.long, device=dev)
        e = unet(torch.cat([inp, inp]), tt, ctx, cross_attention_kwargs={"scale": scale2}).sample.float()
        return e[:B] + guidance_scale * (e[B:] - e[:B])

    x = latents.float() * math.sqrt(float(sig[0]) ** 2 + 1.0)   # init_noise_sigma of the "leading" spacing
    noise_fn = None
    if sampler == "kdpm2a":   # fresh noise per ancestral step from the caller's generator (the reference seeds one per image)
        noise_fn = lambda i, x_: torch.randn(x_.shape, generator=generator, device=x_.device, dtype=x_.dtype)   # noqa: E731
    return k_sample_core(eps_fn, x, ts, sig, sampler, noise_fn=noise_fn)
