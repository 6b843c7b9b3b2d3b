"""Sigma-space ("k-diffusion") samplers of evaluation/utils_eval.py:83-101 on the HIP U-Net: ``euler`` (EulerDiscreteScheduler),
``heun`` (HeunDiscreteScheduler), ``kdpm2`` (KDPM2DiscreteScheduler) and ``lms`` (LMSDiscreteScheduler), restated from the
published algorithms (Karras et al. 2022, Alg. 1 / k-diffusion ``sample_euler``, ``sample_heun``, ``sample_dpm_2``, ``sample_lms``) with
the conventions diffusers applies to the SD-1.5 scheduler config: ``timestep_spacing="leading"``, ``steps_offset=1``, epsilon
prediction, sigma_t = sqrt((1 - acp_t) / acp_t), a final sigma of 0, ``init_noise_sigma = sqrt(sigma_max^2 + 1)``, and
``scale_model_input``: the U-Net sees x / sqrt(sigma^2 + 1).  diffusers is not on disk: UNPINNED (the property tests in
tests/test_samplers.py hold for any consistent solver; the step coefficients are checked against an independent restatement in
oracle/ppft_oracle.py).  The remaining samplers of that table (pndm, dpms_s, dpms_sde, kdpm2a, unipc) are not built.

Host logic in float64, the state x in fp32 on the device; the update between two U-Net calls is a handful of element-wise
torch ops on a [B,4,64,64] tensor (plumbing next to a 5 ms U-Net call).  DDIM and DPM-Solver++(2M) live in inference.py.
"""
import math

import torch

from .inference import _cfg_scale, ddim_timesteps
from .watermark import sd15_alphas_cumprod

SAMPLERS = ("euler", "heun", "kdpm2", "lms")


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


def k_sample_core(eps_fn, x, timesteps, sigmas, sampler="euler", lms_order=4):
    """Integrate dx/dsigma = eps(x, sigma) from sigmas[0] to 0.  ``eps_fn(x, sigma, t)`` returns the (guided) noise prediction for
    the UN-scaled state x at noise level sigma / fractional timestep t.  x: the state in k-space (x0 + sigma * noise)."""
    if sampler not in SAMPLERS:
        raise ValueError(f"sampler {sampler!r} is not one of {SAMPLERS}")
    table = k_sigma_table() if sampler == "kdpm2" else None
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


@torch.no_grad()
def k_sample(unet, ctx_cond, ctx_uncond, latents, sampler="euler", num_inference_steps=50, guidance_scale=7.5, scale=None):
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
    return k_sample_core(eps_fn, x, ts, sig, sampler)
