"""Samplers in front of evaluation / rob-finetune: DDIM (utils_eval.py:83-126) and DPM-Solver++(2M)
(rob_enhance_finetune.py:993,1012).  diffusers is not on disk (UNPINNED); what can be pinned are the solvers' own
invariants: with an EXACT noise model eps(x,t) = (x - alpha_t x0*) / sigma_t every consistent solver must return x0* at the
end of the trajectory, whatever the step count, and the 2M coefficients must equal an independent restatement of the
published update."""
import math

import numpy as np
import pytest
import torch

from aqualora_amd.inference import dpmpp2m_schedule
from oracle import ppft_oracle as O


def test_dpmpp2m_schedule_matches_independent_restatement():
    acp = O.alphas_cumprod().double().numpy()
    for steps in (20, 10, 50):
        sched = dpmpp2m_schedule(steps)
        ts = [r[0] for r in sched]
        want_ts = list(np.linspace(0, 999, steps + 1).round()[::-1][:-1].astype(int))
        assert ts == want_ts and ts[0] == 999 and len(ts) == steps
        lam = lambda t: 0.5 * math.log(acp[t] / (1 - acp[t]))  # noqa: E731
        for i, (t, al, sg, a, b, c) in enumerate(sched):
            nxt = ts[i + 1] if i + 1 < steps else 0
            assert abs(al - math.sqrt(acp[t])) < 1e-12 and abs(sg - math.sqrt(1 - acp[t])) < 1e-12
            h = lam(nxt) - lam(t)
            E = -math.sqrt(acp[nxt]) * math.expm1(-h)
            assert abs(a - math.sqrt((1 - acp[nxt]) / (1 - acp[t]))) < 1e-9
            if i == 0 or (i == steps - 1 and steps < 15):
                assert c == 0.0 and abs(b - E) < 1e-9
            else:
                r0 = (lam(t) - lam(ts[i - 1])) / h
                assert abs(c + 0.5 * E / r0) < 1e-9 and abs(b - (E + 0.5 * E / r0)) < 1e-9


def test_exact_noise_model_is_integrated_exactly():
    torch.manual_seed(0)
    x0_star = torch.randn(2, 4, 8, 8)
    xT = torch.randn(2, 4, 8, 8)
    acp = O.alphas_cumprod().double()

    def eps_model(x, t):
        e = (x.double() - acp[t].sqrt() * x0_star.double()) / (1 - acp[t]).sqrt()
        return e.float(), e.float()
    for steps in (20, 8):
        out = O.dpmpp2m_sample(eps_model, xT, steps, guidance=7.5)   # the oracle's own trajectory
        out_rows = O.dpmpp2m_sample(eps_model, xT, dpmpp2m_schedule(steps), guidance=7.5)   # the product's coefficient rows
        assert float((out - out_rows).abs().max()) < 1e-4 * float(out.abs().max())
        # the trajectory ends at alphas_cumprod[0]: x = alpha_0 x0* + sigma_0 eps_last; x0* is recovered up to that residual
        a0, s0 = float(acp[0].sqrt()), float((1 - acp[0]).sqrt())
        assert float((out / a0 - x0_star).abs().max()) < 4 * s0 / a0 * float(xT.abs().max()) + 1e-3
    # DDIM under the same model
    x = xT.clone()
    ts = [t for t in range(981, 0, -20)]
    for t in ts:
        eu, ec = eps_model(x, t)
        x = O.ddim_step(x, eu, ec, t, t - 20, 7.5)
    assert float((x / float(acp[0].sqrt()) - x0_star).abs().max()) < 0.2


@pytest.mark.gpu
def test_dpm_solver_hip_vs_oracle_tiny_unet():
    """The graph-replayed HIP sampler against the oracle loop driving the same tiny U-Net (fp32 update kernel, bf16 U-Net)."""
    from aqualora_amd.inference import dpm_solver_sample
    from tests.common import T, TINY, tiny_unet
    dev = "cuda"
    unet = tiny_unet(dev, torch.bfloat16)
    ctx = T("s.ctx", (1, 77, TINY["cross_attention_dim"]), device=dev)
    unc = torch.zeros_like(ctx)
    lat = T("s.lat", (1, 4, 16, 16), device=dev)
    got = dpm_solver_sample(unet, ctx, unc, lat, 6, 3.0)
    got_eager = dpm_solver_sample(unet, ctx, unc, lat, 6, 3.0, graph=False)

    def eps_model(x, t):
        tt = torch.full((2,), t, dtype=torch.long, device=dev)
        e = unet(torch.cat([x.to(dev), x.to(dev)]), tt, torch.cat([unc, ctx]).to(torch.bfloat16),
                 cross_attention_kwargs={"scale": None}).sample.float().cpu()
        return e[:1], e[1:]
    with torch.no_grad():
        want = O.dpmpp2m_sample(eps_model, lat.cpu(), 6, guidance=3.0)   # trajectory built by the oracle itself
    assert torch.isfinite(got).all()
    assert float((got.cpu() - want).abs().max() / want.abs().max()) < 2e-2
    assert float((got - got_eager).abs().max() / got_eager.abs().max()) < 1e-5

@pytest.mark.parametrize("sampler", ["euler", "heun", "kdpm2", "lms"])
def test_k_samplers_integrate_the_exact_noise_model_and_match_the_restatement(sampler):
    """Sigma-space samplers (evaluation/utils_eval.py:83-101).  (1) With the exact noise model eps(x, sigma) = (x - x0) / sigma the
    probability-flow ODE is linear in sigma and every consistent solver must land on x0 at sigma = 0.  (2) With a nonlinear
    stand-in model the host loop equals the independent restatement in oracle/ppft_oracle.py step for step (1e-9, float64)."""
    from aqualora_amd.ksamplers import k_sample_core, k_schedule, lms_coefficient, sigma_to_t, k_sigma_table
    from aqualora_amd.watermark import sd15_alphas_cumprod
    torch.manual_seed(0)
    ts, sig = k_schedule(12)
    ts_o, sig_o = O.k_sigmas_oracle(12, sd15_alphas_cumprod().double())
    assert ts == ts_o and float((sig - torch.tensor(sig_o, dtype=torch.float64)).abs().max()) < 1e-12
    x0 = torch.randn(2, 4, 8, 8, dtype=torch.float64)
    noise = torch.randn_like(x0)
    x = x0 + float(sig[0]) * noise
    got = k_sample_core(lambda x_, s, t: (x_ - x0) / s, x, ts, sig, sampler)
    assert float((got - x0).abs().max()) < 1e-9
    # nonlinear model: the two implementations must walk the same trajectory
    f = lambda x_, s, t: torch.tanh(x_ / (1 + s)) * (1 + 0.1 * s)   # noqa: E731
    a = k_sample_core(f, x, ts, sig, sampler)
    b = O.k_sample_oracle(f, x, ts_o, sig_o, sampler)
    assert float((a - b).abs().max() / b.abs().max()) < 1e-9
    if sampler == "lms":
        for i, order in ((0, 1), (1, 2), (5, 4)):
            for j in range(order):
                assert abs(lms_coefficient(sig, order, i, j) - O.lms_coeff_oracle(sig_o, order, i, j)) < 1e-9
    if sampler == "kdpm2":   # the fractional timestep of a table sigma is its index; midpoints fall strictly between
        tab = k_sigma_table()
        assert abs(sigma_to_t(tab[417], tab) - 417.0) < 1e-6
        assert 400.0 < sigma_to_t(math.sqrt(float(tab[400]) * float(tab[401])), tab) < 401.0


@pytest.mark.gpu
def test_k_sampler_hip_vs_oracle_tiny_unet():
    """euler and heun on the HIP tiny U-Net (fractional-free timesteps) against the oracle loop driving the same U-Net."""
    from aqualora_amd.ksamplers import k_sample, k_schedule
    from tests.common import T, TINY, tiny_unet
    dev = "cuda"
    unet = tiny_unet(dev, torch.bfloat16)
    ctx = T("k.ctx", (1, 77, TINY["cross_attention_dim"]), device=dev)
    unc = torch.zeros_like(ctx)
    lat = T("k.lat", (1, 4, 16, 16), device=dev)
    ts, sig = k_schedule(5)
    for sampler in ("euler", "heun", "kdpm2"):
        got = k_sample(unet, ctx, unc, lat, sampler, 5, 3.0)

        def eps(x, s, t):
            if t is None:
                from aqualora_amd.ksamplers import sigma_to_t
                t = sigma_to_t(s)
            inp = (x / math.sqrt(s * s + 1)).to(dev).float()
            tt = torch.full((2,), float(t), dtype=torch.long if float(t) == int(t) else torch.float32, device=dev)
            e = unet(torch.cat([inp, inp]), tt, torch.cat([unc, ctx]).to(torch.bfloat16),
                     cross_attention_kwargs={"scale": None}).sample.float().cpu()
            return e[:1] + 3.0 * (e[1:] - e[:1])
        with torch.no_grad():
            want = O.k_sample_oracle(eps, lat.cpu().float() * math.sqrt(float(sig[0]) ** 2 + 1), ts, [float(v) for v in sig], sampler)
        assert torch.isfinite(got).all()
        # bf16 U-Net inputs: an fp32 ulp between the host-side and device-side update flips input roundings (the DPM-Solver test's bound)
        assert float((got.cpu() - want).abs().max() / want.abs().max()) < 2e-2, sampler


def test_pndm_plms_integrates_the_exact_noise_model_and_matches_the_restatement():
    """``pndm`` (evaluation/utils_eval.py:91-92; PNDMScheduler with skip_prk_steps = PLMS).  (1) Under the exact noise model the true
    noise is constant along the trajectory, every slope combination with weights summing to one equals it, and the transfer formula
    is exact: the loop must land on alpha_0 x0 + sigma_0 eps to machine precision for any step count.  (2) With a nonlinear stand-in
    model the scheduler-style counter loop equals the phase-by-phase restatement in oracle/ppft_oracle.py.  (3) The timestep list has
    the warm-up repeat diffusers builds."""
    from aqualora_amd.ksamplers import pndm_sample_core, pndm_timesteps
    torch.manual_seed(0)
    acp = O.alphas_cumprod().double()
    assert pndm_timesteps(10) == ([901, 801, 801, 701, 601, 501, 401, 301, 201, 101, 1], 100)
    ts50, r50 = pndm_timesteps(50)
    assert len(ts50) == 51 and ts50[:3] == [981, 961, 961] and ts50[-1] == 1 and r50 == 20
    x0 = torch.randn(2, 4, 8, 8, dtype=torch.float64)
    eps = torch.randn_like(x0)
    for steps in (10, 25, 50):
        t0 = pndm_timesteps(steps)[0][0]
        xT = acp[t0].sqrt() * x0 + (1 - acp[t0]).sqrt() * eps
        got = pndm_sample_core(lambda x, t: (x - acp[t].sqrt() * x0) / (1 - acp[t]).sqrt(), xT, steps, acp)
        assert float((got - (acp[0].sqrt() * x0 + (1 - acp[0]).sqrt() * eps)).abs().max()) < 1e-9
        f = lambda x, t: torch.tanh(x * 0.7) * (1 + 0.001 * t)   # noqa: E731
        a, b = pndm_sample_core(f, xT, steps, acp), O.plms_oracle(f, xT, steps, acp)
        assert float((a - b).abs().max() / b.abs().max()) < 1e-9


def test_kdpm2_ancestral_matches_the_restatement_and_keeps_the_noise_level():
    """``kdpm2a`` (utils_eval.py:99-100): same trajectory as the independent restatement given the same per-step noise; sigma_down^2 +
    sigma_up^2 = sigma_next^2 (the marginal noise level of an ancestral step); with zero fresh noise and the exact model the state
    after a step is x0 + sigma_down * noise."""
    from aqualora_amd.ksamplers import ancestral_step, k_sample_core, k_schedule
    from aqualora_amd.watermark import sd15_alphas_cumprod
    torch.manual_seed(1)
    ts, sig = k_schedule(12)
    ts_o, sig_o = O.k_sigmas_oracle(12, sd15_alphas_cumprod().double())
    for i in range(12):
        d, u = ancestral_step(float(sig[i]), float(sig[i + 1]))
        assert abs(d * d + u * u - float(sig[i + 1]) ** 2) < 1e-12 * max(1.0, float(sig[i + 1]) ** 2) and 0 <= d <= float(sig[i + 1])
    x0 = torch.randn(2, 4, 8, 8, dtype=torch.float64)
    n0 = torch.randn_like(x0)
    x = x0 + float(sig[0]) * n0
    g = torch.Generator().manual_seed(2)
    noises = [torch.randn(2, 4, 8, 8, dtype=torch.float64, generator=g) for _ in range(12)]
    f = lambda x_, s, t: torch.tanh(x_ / (1 + s)) * (1 + 0.1 * s)   # noqa: E731
    a = k_sample_core(f, x, ts, sig, "kdpm2a", noise_fn=lambda i, x_: noises[i])
    f.noise = lambda i, x_: noises[i]
    b = O.k_sample_oracle(f, x, ts_o, sig_o, "kdpm2a")
    assert float((a - b).abs().max() / b.abs().max()) < 1e-9
    one = k_sample_core(lambda x_, s, t: (x_ - x0) / s, x, ts[:1], sig[:2], "kdpm2a", noise_fn=lambda i, x_: torch.zeros_like(x_))
    assert float((one - (x0 + ancestral_step(float(sig[0]), float(sig[1]))[0] * n0)).abs().max()) < 1e-9
    with pytest.raises(ValueError):
        k_sample_core(f, x, ts, sig, "kdpm2a")


@pytest.mark.gpu
def test_pndm_hip_vs_oracle_tiny_unet():
    """PLMS on the HIP tiny U-Net against the oracle's phase-by-phase loop driving the same U-Net."""
    from aqualora_amd.ksamplers import pndm_sample
    from tests.common import T, TINY, tiny_unet
    dev = "cuda"
    unet = tiny_unet(dev, torch.bfloat16)
    ctx = T("p.ctx", (1, 77, TINY["cross_attention_dim"]), device=dev)
    unc = torch.zeros_like(ctx)
    lat = T("p.lat", (1, 4, 16, 16), device=dev)
    got = pndm_sample(unet, ctx, unc, lat, 6, 3.0)

    def eps(x, t):
        tt = torch.full((2,), int(t), dtype=torch.long, device=dev)
        xin = x.to(dev).float().contiguous()
        e = unet(torch.cat([xin, xin]), tt, torch.cat([unc, ctx]).to(torch.bfloat16),
                 cross_attention_kwargs={"scale": None}).sample.float().cpu()
        return e[:1] + 3.0 * (e[1:] - e[:1])
    with torch.no_grad():
        want = O.plms_oracle(eps, lat.cpu().float(), 6, O.alphas_cumprod().double())
    assert torch.isfinite(got).all()
    assert float((got.cpu() - want).abs().max() / want.abs().max()) < 2e-2
