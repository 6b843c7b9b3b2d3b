"""GPU parity tests (MI355X): HIP path through the C-ABI vs the CPU oracle and the reference-generated golden vectors.

Tolerances (bf16 storage, fp32 accumulation; the oracle mirrors the bf16 rounding points):
  single op vs fp32 golden ............ 1.5e-2 of the tensor's max (bf16 has 8 mantissa bits)
  tiny U-Net output vs bf16 oracle ..... 3e-2 ; vs fp32 golden 5e-2
  LoRA gradient norms .................. 5e-2 relative
"""
import numpy as np
import pytest
import torch

from tests.common import LORA_CASES, SEED, T, TINY, TINY_RANK, ppft_inputs, tiny_lora, tiny_unet

pytestmark = pytest.mark.gpu
DEV = "cuda"


def relerr(a, b):
    a = torch.as_tensor(np.asarray(a) if not torch.is_tensor(a) else a).detach().double().cpu()
    b = torch.as_tensor(np.asarray(b) if not torch.is_tensor(b) else b).detach().double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def l2rel(a, b):
    a, b = a.detach().double().cpu().flatten(), b.detach().double().cpu().flatten()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def test_library_is_loaded_and_complete():
    from aqualora_amd import _lib
    lib = _lib.load()
    for name in _lib.SIGNATURES:
        assert hasattr(lib, name)


def test_lora_forwards_vs_reference_golden(golden):
    from aqualora_amd import lora as AL
    g = golden("lora_forwards.npz")
    for tag, cin, cout, n, r in LORA_CASES:
        host = AL.LoRACompatibleLinear(cin, cout, device=DEV, dtype=torch.bfloat16)
        ll = AL.LoRALinearLayer(cin, cout, r, device=DEV, dtype=torch.float32)
        with torch.no_grad():
            host.weight.copy_(T(f"{tag}.w", (cout, cin), cin ** -0.5))
            host.bias.copy_(T(f"{tag}.b", (cout,), 0.02))
            ll.down.weight.copy_(T(f"{tag}.down", (r, cin), 1.0 / r))
            ll.up.weight.copy_(T(f"{tag}.up", (cout, r), 0.05))
        host.set_lora_layer(ll)
        x = T(f"{tag}.x", (2, n, cin), device=DEV).to(torch.bfloat16).requires_grad_(True)
        S = (T(f"{tag}.S", (2, r), 0.3, DEV) + 1.0).requires_grad_(True)
        y = AL.CustomLoRACompatibleLinearforward(host, x, S)
        y.backward(T(f"{tag}.dy", (2, n, cout), device=DEV).to(torch.bfloat16))
        assert relerr(y, g[f"{tag}.y"]) < 1.5e-2
        assert relerr(x.grad, g[f"{tag}.dx"]) < 1.5e-2
        assert relerr(S.grad, g[f"{tag}.dS"]) < 2e-2
        assert relerr(ll.down.weight.grad, g[f"{tag}.ddown"]) < 2e-2
        assert relerr(ll.up.weight.grad, g[f"{tag}.dup"]) < 2e-2
        assert relerr(AL.CustomLoRACompatibleLinearforward(host, x.detach(), 0.5), g[f"{tag}.y_float_scale"]) < 1.5e-2
        assert relerr(AL.CustomLoRALinearLayerforward(ll, x.detach(), S.detach()), g[f"{tag}.lora_only"]) < 2e-2
        ll.network_alpha = r / 4.0      # `up_hidden_states *= network_alpha / rank` (lora_modules.py:21-22): a quarter of the branch
        assert relerr(AL.CustomLoRALinearLayerforward(ll, x.detach(), S.detach()), 0.25 * g[f"{tag}.lora_only"]) < 2e-2
        ll.network_alpha = None
        if f"{tag}.conv_y" in g:
            hc = AL.LoRACompatibleConv(cin, cout, 1, device=DEV, dtype=torch.bfloat16)
            lc = AL.LoRAConv2dLayer(cin, cout, r).to(DEV)
            with torch.no_grad():
                hc.weight.copy_(host.weight.view(cout, cin, 1, 1)); hc.bias.copy_(host.bias)
                lc.down.weight.copy_(ll.down.weight.view(r, cin, 1, 1)); lc.up.weight.copy_(ll.up.weight.view(cout, r, 1, 1))
            hc.set_lora_layer(lc)
            xc = x.detach().permute(0, 2, 1).reshape(2, cin, 4, n // 4)
            assert relerr(AL.CustomLoRACompatibleConvforward(hc, xc, S.detach()), g[f"{tag}.conv_y"]) < 1.5e-2
            assert relerr(AL.CustomLoRAConv2dLayerforward(lc, xc, S.detach()),
                          g[f"{tag}.conv_y"] - torch.nn.functional.conv2d(
                              xc.float().cpu(), hc.weight.detach().float().cpu(),
                              hc.bias.detach().float().cpu()).numpy()) < 5e-2


def test_watermark_modules_vs_reference_golden(golden):
    from aqualora_amd import synth
    from aqualora_amd.watermark import MapperNet, SecretEncoder, customDDPMScheduler
    from oracle import ppft_oracle as O
    g = golden("watermark.npz")
    mp = MapperNet(48, 32).to(DEV)
    with torch.no_grad():
        mp.bit_embeddings.weight.copy_(T("mapper.E", (48, 32)))
    msg = synth.bits("msg", (4, 48), SEED, DEV)
    S = mp(msg)
    S.backward(T("mapper.dS", (4, 32), device=DEV))
    assert relerr(S, g["S"]) < 1e-5 and relerr(mp.bit_embeddings.weight.grad, g["dE"]) < 1e-5
    enc = SecretEncoder(48).to(DEV)
    assert enc.encode(msg).abs().max().item() == 0.0  # zero-init invariant (models.py:63)
    with torch.no_grad():
        enc.secret_scaler[0].weight.copy_(T("enc.lin.w", (1024, 48), 48 ** -0.5))
        enc.secret_scaler[0].bias.copy_(T("enc.lin.b", (1024,), 0.1))
        enc.secret_scaler[5].weight.copy_(T("enc.conv.w", (4, 4, 3, 3), 0.05))
        enc.secret_scaler[5].bias.copy_(T("enc.conv.b", (4,), 0.01))
    x = T("enc.x", (4, 4, 64, 64), device=DEV)
    xc, c = enc(x, msg)
    assert relerr(c, g["c"]) < 1e-5
    assert abs(xc.double().sum().item() - float(g["x_plus_c_checksum"])) < 1e-2
    sch = customDDPMScheduler(device=DEV)
    z, e = T("an.x", (4, 4, 64, 64), device=DEV), T("an.n", (4, 4, 64, 64), device=DEV)
    t = torch.tensor([0, 1, 500, 999], device=DEV)
    a, b = sch.add_noise_pair(z, c, e, t)
    # the kernel computes the fp32 closed form and stores bf16 (the reference casts to weight_dtype): compare with the
    # bf16 rounding of the oracle's fp32 result ELEMENT BY ELEMENT -- equal bits except where the two fp32 values straddle
    # a rounding boundary (then exactly one bf16 ulp apart)
    for got, want32 in ((a, O.add_noise(z.cpu(), e.cpu(), t.cpu())), (b, O.add_noise((z + c).cpu(), e.cpu(), t.cpu()))):
        want = want32.to(torch.bfloat16)
        got = got.cpu()
        off = got != want
        assert off.float().mean().item() < 2e-3, off.float().mean().item()
        # half a bf16 ulp of rounding + the fp32 cancellation noise of the two products (fma contraction vs two roundings)
        slack = (got.float() - want32).abs() - (2.0 ** -8 * want32.abs() + 1e-6)
        assert slack.max().item() <= 0.0, slack.max().item()
        assert l2rel(got.float(), want32) < 3e-3


def _gpu_tiny(rank=TINY_RANK, up_std=0.1):
    from aqualora_amd.lora import inject_lora, patch_lora_forwards
    from aqualora_amd.unet import lora_keys
    unet = tiny_unet(DEV, torch.bfloat16)
    keys = lora_keys(unet)
    lw = tiny_lora(keys, unet, rank, up_std)
    state = {}
    for k, (d, u) in lw.items():
        state[k + ".down.weight"], state[k + ".up.weight"] = d, u
    inject_lora(unet, rank, keys, state)
    patch_lora_forwards(unet)
    return unet, keys, lw


def test_network_alpha_reaches_every_site_of_the_unet():
    """`up_hidden_states *= network_alpha / rank` (utils/lora_modules.py:21-22, 39-40) at U-Net level, rank 32 with a LoraBank
    (the configuration in which q|k|v, the text-state k|v and the feed-forward take the grouped / fused launches): a model
    whose 192 LoRA layers carry network_alpha = 8 under scale S must equal the same model without alpha under S * 8/32 --
    forward and the gradient that reaches S -- i.e. no site drops or double-applies the factor."""
    from aqualora_amd.lora import LoraBank, inject_lora, patch_lora_forwards
    from aqualora_amd.unet import lora_keys
    rank = 32
    # channel counts that are multiples of 160 (head dims 80 / 160): the shapes the grouped launches accept
    cfg = dict(block_out_channels=(160, 320, 320, 320), cross_attention_dim=32, attention_heads=2, layers_per_block=1)
    unet = tiny_unet(DEV, torch.bfloat16, cfg)
    keys = lora_keys(unet)
    state = {}
    for k, (d, u) in tiny_lora(keys, unet, rank, 0.1).items():
        state[k + ".down.weight"], state[k + ".up.weight"] = d, u
    inject_lora(unet, rank, keys, state)
    patch_lora_forwards(unet)
    bank = LoraBank(unet)      # stacks the bf16 A / Bup copies: the grouped launches become eligible
    inp = ppft_inputs(cfg, rank=rank, device=DEV)
    x, t, ctx = inp["z"].to(torch.bfloat16), inp["t"], inp["ctx"].to(torch.bfloat16)
    S0 = 1.0 + 0.3 * T("alpha.S", (x.shape[0], rank), 1.0, DEV)

    def run(alpha, S):
        for k in keys:
            unet.get_submodule(k).lora_layer.network_alpha = alpha
        S = S.clone().requires_grad_(True)
        y = unet(x, t, ctx, cross_attention_kwargs={"scale": S}).sample
        y.float().square().mean().backward()
        return y.detach(), S.grad.detach()

    from aqualora_amd import ops
    calls = []
    real = ops.lora_linear_grouped
    ops.lora_linear_grouped = lambda *a: (calls.append(len(a[3])), real(*a))[1]
    try:
        y_ref, g_ref = run(None, S0 * (8.0 / rank))
        assert calls and max(calls) > 3, calls      # the alpha-free model does take the grouped q|k|v and text k|v launches
        calls.clear()
        y_a, g_a = run(8.0, S0)
        assert not calls                            # ... and a model with network_alpha does not
    finally:
        ops.lora_linear_grouped = real
    y_none, _ = run(None, S0)
    assert relerr(y_a, y_ref) < 1e-2, relerr(y_a, y_ref)
    assert relerr(y_none, y_ref) > 4 * relerr(y_a, y_ref) + 1e-3   # the factor matters here: a dropped alpha would be seen
    # dL/dS carries the chain-rule factor alpha / rank
    assert l2rel(g_a, g_ref * (8.0 / rank)) < 3e-2, l2rel(g_a, g_ref * (8.0 / rank))


def test_tiny_unet_forward_vs_oracle_and_golden(golden):
    from oracle import ppft_oracle as O
    g = golden("tiny_ppft.npz")
    unet, keys, lw = _gpu_tiny()
    inp = ppft_inputs()
    acp = O.alphas_cumprod()
    x_t = O.add_noise(inp["z"], inp["eps"], inp["t"], acp)
    x_wm = O.add_noise(inp["z"] + inp["wm"], inp["eps"], inp["t"], acp)
    S = O.mapper(inp["msg"], inp["E"])
    ref = O.UNetOracle(tiny_unet().state_dict(), TINY, lw, bf16=True)
    with torch.no_grad():
        clean_o = ref.forward(x_t, inp["t"], inp["ctx"], None)
        pred_o = ref.forward(x_wm, inp["t"], inp["ctx"], S)
        clean = unet(x_t.to(DEV), inp["t"].to(DEV), inp["ctx"].to(DEV), cross_attention_kwargs={"scale": None}).sample
        clean0 = unet(x_t.to(DEV), inp["t"].to(DEV), inp["ctx"].to(DEV),
                      cross_attention_kwargs={"scale": torch.zeros_like(S).to(DEV)}).sample
        pred = unet(x_wm.to(DEV), inp["t"].to(DEV), inp["ctx"].to(DEV), cross_attention_kwargs={"scale": S.to(DEV)}).sample
    assert torch.equal(clean, clean0)  # skipping the LoRA branch == the reference's zero scale, bit for bit
    assert relerr(clean, clean_o) < 3e-2 and relerr(pred, pred_o) < 3e-2
    assert relerr(clean, g["clean"]) < 5e-2 and relerr(pred, g["pred"]) < 5e-2


def test_tiny_unet_large_non_square_latents_vs_oracle():
    """rob-finetune samples at 512..768 px, height and width drawn independently (rob_enhance_finetune.py:1004-1005): the U-Net
    runs on non-square latents up to 96x96.  Tiny U-Net, 72x88 latents (576x704 px; 9x11 at the lowest level), per-sample
    scale rows, forward against the bf16-mirroring oracle."""
    from oracle import ppft_oracle as O
    unet, keys, lw = _gpu_tiny()
    B, H, W = 2, 72, 88
    x = T("ns.x", (B, 4, H, W))
    ctx = T("ns.ctx", (B, 77, TINY["cross_attention_dim"]))
    t = torch.tensor([40, 900])
    S = 1.0 + 0.3 * T("ns.S", (B, TINY_RANK))
    ref = O.UNetOracle(tiny_unet().state_dict(), TINY, lw, bf16=True)
    with torch.no_grad():
        want = ref.forward(x, t, ctx, S)
        got = unet(x.to(DEV), t.to(DEV), ctx.to(DEV).to(torch.bfloat16), cross_attention_kwargs={"scale": S.to(DEV)}).sample
    assert got.shape == (B, 4, H, W)
    assert relerr(got, want) < 3e-2 and l2rel(got, want) < 2e-2, (relerr(got, want), l2rel(got, want))


def test_full_size_sd15_unet_forward_vs_oracle():
    """BASELINE size: the full SD-1.5 U-Net (859.5 M synthetic parameters, 64x64x4 latents, 77x768 context) with the
    rank-32 watermark LoRA on all 192 sites -- HIP forward (clean and watermarked) against the bf16-mirroring CPU oracle.
    Also the size-independent properties: zero scale == skipped LoRA branch bit for bit, run-to-run determinism, and the
    LoRA branch changing the prediction."""
    from aqualora_amd import synth
    from aqualora_amd.lora import inject_lora
    from aqualora_amd.unet import SD15, UNet2DConditionModel, init_synthetic, lora_keys
    from oracle import ppft_oracle as O
    rank, seed = 32, 2048
    unet = UNet2DConditionModel(device=DEV, dtype=torch.bfloat16)
    init_synthetic(unet, seed)
    keys = lora_keys(unet)
    assert len(keys) == 192
    sd = {k: v.detach().float().cpu() for k, v in unet.state_dict().items()}
    inject_lora(unet, rank, keys)
    lw = {}
    with torch.no_grad():
        for k in keys:
            lay = unet.get_submodule(k).lora_layer
            lay.down.weight.copy_(synth.normal(k + ".lora.down", lay.down.weight.shape, 1.0 / rank, seed, DEV))
            lay.up.weight.copy_(synth.normal(k + ".lora.up", lay.up.weight.shape, 0.02, seed, DEV))
            lw[k] = (lay.down.weight.detach().float().cpu(), lay.up.weight.detach().float().cpu())
    x = synth.normal("full.z", (1, 4, 64, 64), 1.0, seed)
    ctx = synth.normal("full.ctx", (1, 77, 768), 1.0, seed)
    t = torch.tensor([500])
    S = 1.0 + 0.5 * synth.normal("full.S", (1, rank), 1.0, seed)
    torch.set_num_threads(min(32, len(__import__("os").sched_getaffinity(0))))
    ref = O.UNetOracle(sd, dict(SD15), lw, bf16=True)
    with torch.no_grad():
        clean_o = ref.forward(x, t, ctx, None)
        pred_o = ref.forward(x, t, ctx, S)
        xb, cb = x.to(DEV), ctx.to(DEV).to(torch.bfloat16)
        clean = unet(xb, t.to(DEV), cb, cross_attention_kwargs={"scale": None}).sample
        clean0 = unet(xb, t.to(DEV), cb, cross_attention_kwargs={"scale": torch.zeros_like(S).to(DEV)}).sample
        pred = unet(xb, t.to(DEV), cb, cross_attention_kwargs={"scale": S.to(DEV)}).sample
        pred2 = unet(xb, t.to(DEV), cb, cross_attention_kwargs={"scale": S.to(DEV)}).sample
    assert torch.equal(clean, clean0) and torch.equal(pred, pred2)
    assert relerr(clean, clean_o) < 4e-2, relerr(clean, clean_o)
    assert relerr(pred, pred_o) < 4e-2, relerr(pred, pred_o)
    assert l2rel(clean, clean_o) < 2e-2 and l2rel(pred, pred_o) < 2e-2, (l2rel(clean, clean_o), l2rel(pred, pred_o))
    lora_effect = relerr(pred_o, clean_o)
    eff = l2rel(pred.float() - clean.float(), pred_o - clean_o)
    print(f"full-size forward: clean {relerr(clean, clean_o):.3e} pred {relerr(pred, pred_o):.3e} "
          f"LoRA effect {lora_effect:.3e}, effect l2rel {eff:.3e}")
    assert lora_effect > 1e-3 and eff < 0.1, (lora_effect, eff)


@pytest.mark.parametrize("rank,force_chains", [pytest.param(32, False, id="32"), pytest.param(320, False, id="320"),
                                               pytest.param(320, True, id="320-chains")])
def test_full_size_ppft_gradients_vs_oracle(rank, force_chains):
    """BASELINE sizes (config 2: rank 32; configs 3 / 5: rank 320): ONE full PPFT step at batch 1 on the full SD-1.5 U-Net --
    clean forward, watermarked forward, MSE, backward to all 384 LoRA tensors + the mapper -- HIP against the
    bf16-mirroring CPU oracle (oracle.ppft_loss).  Checked: loss, predicted noise (max-norm AND L2), the gradient norm of
    every one of the 384 tensors, and every full gradient tensor in relative L2 (worst / median stated below).
    ``320-chains`` (round 6): the chains' tile-count gate lowered (ops.CHAIN_MIN_TILES) so that at batch 1 the
    320-channel level runs on the rank-320 row-resident chain kernel (aql_lora_chain_fwd_r320, 15 launches) and its q | k | v stages on the
    grouped backward -- the kernel meets the oracle DIRECTLY, not only through bit identity with the per-launch path."""
    import os
    from aqualora_amd import ops
    chain_calls = []
    real_chain_fwd, min_tiles = ops.chain_fwd, ops.CHAIN_MIN_TILES
    if force_chains:
        ops.CHAIN_MIN_TILES = 32      # twin batch of 2 x 4096 rows = 128 tiles of 64 >= 2 x 32
        ops.chain_fwd = lambda *a, **kw: (chain_calls.append(kw.get("rank", a[7] if len(a) > 7 else 32)), real_chain_fwd(*a, **kw))[1]
    try:
        _full_size_gradients_vs_oracle(rank)
    finally:
        ops.chain_fwd, ops.CHAIN_MIN_TILES = real_chain_fwd, min_tiles
    if force_chains:
        assert len(chain_calls) == 15 and all(c == 320 for c in chain_calls), chain_calls


def _full_size_gradients_vs_oracle(rank):
    import os
    from aqualora_amd import synth
    from aqualora_amd.lora import inject_lora
    from aqualora_amd.ppft import PPFTTrainer
    from aqualora_amd.unet import SD15, UNet2DConditionModel, init_synthetic, lora_keys
    from aqualora_amd.watermark import MapperNet, SecretEncoder
    from oracle import ppft_oracle as O
    seed = 2048
    unet = UNet2DConditionModel(device=DEV, dtype=torch.bfloat16)
    init_synthetic(unet, seed)
    keys = lora_keys(unet)
    sd = {k: v.detach().float().cpu() for k, v in unet.state_dict().items()}
    inject_lora(unet, rank, keys)
    lo = {}
    with torch.no_grad():
        for k in keys:
            lay = unet.get_submodule(k).lora_layer
            lay.down.weight.copy_(synth.normal(k + ".lora.down", lay.down.weight.shape, 1.0 / rank, seed, DEV))
            lay.up.weight.copy_(synth.normal(k + ".lora.up", lay.up.weight.shape, 0.02, seed, DEV))
            lo[k] = (lay.down.weight.detach().float().cpu().clone().requires_grad_(True),
                     lay.up.weight.detach().float().cpu().clone().requires_grad_(True))
    mapper = MapperNet(48, rank)
    E = synth.normal("fullg.E", (48, rank), 1.0, seed)
    with torch.no_grad():
        mapper.bit_embeddings.weight.copy_(E)
    tr = PPFTTrainer(unet, mapper, SecretEncoder(48), rank)
    z = synth.normal("fullg.z", (1, 4, 64, 64), 1.0, seed)
    wm = synth.normal("fullg.wm", (1, 4, 64, 64), 0.05, seed)
    eps = synth.normal("fullg.eps", (1, 4, 64, 64), 1.0, seed)
    msg = synth.bits("fullg.msg", (1, 48), seed)
    ctx = synth.normal("fullg.ctx", (1, 77, 768), 1.0, seed)
    t = torch.tensor([500])
    tr.sec_encoder.encode = lambda m, out_scale=1.0: wm.to(DEV)
    loss, pred, clean = tr.forward_backward(z.to(DEV), msg.to(DEV), eps.to(DEV), t.to(DEV), ctx.to(DEV).to(torch.bfloat16))
    torch.cuda.synchronize()
    torch.set_num_threads(min(32, len(os.sched_getaffinity(0))))
    Eo = E.clone().requires_grad_(True)
    lo_loss, pred_o, clean_o, _ = O.ppft_loss(sd, dict(SD15), lo, Eo, msg, z, wm, eps, t, ctx, bf16=True)
    lo_loss.backward()
    assert relerr(clean, clean_o) < 4e-2 and relerr(pred, pred_o) < 4e-2
    assert l2rel(clean, clean_o) < 2e-2 and l2rel(pred, pred_o) < 2e-2
    assert abs(loss.item() - lo_loss.item()) < 0.1 * lo_loss.item(), (loss.item(), lo_loss.item())
    gmax = max(p.grad.norm().item() for pr in lo.values() for p in pr)
    rels, norm_err = [], []
    for k in keys:
        lay = unet.get_submodule(k).lora_layer
        for got, want in ((lay.down.weight.grad, lo[k][0].grad), (lay.up.weight.grad, lo[k][1].grad)):
            assert torch.isfinite(got).all()
            wn = want.norm().item()
            if wn > 1e-3 * gmax:      # tensors that carry signal: norm within 10 %, full tensor in relative L2
                norm_err.append(abs(got.norm().item() - wn) / wn)
                rels.append(l2rel(got, want.reshape(got.shape)))
    rels_s = sorted(rels)
    print(f"full-size r={rank} gradients: {len(rels)} tensors, l2rel median {rels_s[len(rels_s) // 2]:.3f} worst "
          f"{rels_s[-1]:.3f}; norm error median {sorted(norm_err)[len(norm_err) // 2]:.3f} worst {max(norm_err):.3f}; "
          f"loss {loss.item():.5e} vs {lo_loss.item():.5e}")
    assert len(rels) >= 300
    # measured on MI355X (round 3): r=32 median 0.022 / worst 0.055 / norm error worst 0.021; r=320 median 0.034 / worst 0.067 /
    # norm error worst 0.015 (bf16 activations + weights).  Bounds = 2x measured, never looser than round 2's 0.06 / 0.12 / 0.05
    b_med, b_worst, b_norm = {32: (0.045, 0.11, 0.042), 320: (0.06, 0.12, 0.035)}[rank]
    assert rels_s[len(rels_s) // 2] < b_med and rels_s[-1] < b_worst, (rels_s[len(rels_s) // 2], rels_s[-1])
    assert max(norm_err) < b_norm, max(norm_err)
    assert l2rel(mapper.bit_embeddings.weight.grad, Eo.grad) < 0.1


def test_full_size_batch4_twin_step_equals_the_mean_of_four_batch1_steps():
    """Shapes that only exist at the benchmark's batch size -- the 12-wave 256x160 conv kernel, the 128-row K-grouped q|k|v
    backward, 8-sample GroupNorm passes, 32768-row twin GEMMs -- are never reached by the batch-1 oracle comparison above.
    Samples are independent, so ONE batch-4 step must equal four batch-1 steps (which ARE pinned to the oracle): the
    predicted noise per sample, the loss (mean of the four), and the flat gradient (LoRA + mapper; mean of the four)."""
    from aqualora_amd import synth
    from aqualora_amd.lora import inject_lora
    from aqualora_amd.ppft import PPFTTrainer
    from aqualora_amd.unet import UNet2DConditionModel, init_synthetic, lora_keys
    from aqualora_amd.watermark import MapperNet, SecretEncoder
    seed, rank, B = 4096, 32, 4
    unet = UNet2DConditionModel(device=DEV, dtype=torch.bfloat16)
    init_synthetic(unet, seed)
    keys = lora_keys(unet)
    inject_lora(unet, rank, keys)
    with torch.no_grad():
        for k in keys:
            lay = unet.get_submodule(k).lora_layer
            lay.down.weight.copy_(synth.normal(k + ".lora.down", lay.down.weight.shape, 1.0 / rank, seed, DEV))
            lay.up.weight.copy_(synth.normal(k + ".lora.up", lay.up.weight.shape, 0.02, seed, DEV))
    mapper = MapperNet(48, rank)
    with torch.no_grad():
        mapper.bit_embeddings.weight.copy_(synth.normal("b4.E", (48, rank), 1.0, seed))
    tr = PPFTTrainer(unet, mapper, SecretEncoder(48), rank)
    z = synth.normal("b4.z", (B, 4, 64, 64), 1.0, seed).to(DEV)
    wm = synth.normal("b4.wm", (B, 4, 64, 64), 0.05, seed).to(DEV)
    eps = synth.normal("b4.eps", (B, 4, 64, 64), 1.0, seed).to(DEV)
    msg = synth.bits("b4.msg", (B, 48), seed).to(DEV)
    ctx = synth.normal("b4.ctx", (B, 77, 768), 1.0, seed).to(DEV).to(torch.bfloat16)
    t = torch.tensor([500, 20, 981, 333], device=DEV)
    cur = {"sl": slice(0, B)}
    tr.sec_encoder.encode = lambda m, out_scale=1.0: wm[cur["sl"]]
    n = tr.bank.numel
    tr.bank.zero_grad()
    loss4, pred4, clean4 = tr.forward_backward(z, msg, eps, t, ctx)
    g4 = tr.bank.grad[:n].clone()
    # Run-to-run spread: dA / dB (gemm_tn_tr_grouped_kernel) and dS (lora_ds) accumulate with fp32 atomics, so the summation
    # order -- and the last bits -- change from run to run; everything upstream is deterministic.  The same step again:
    tr.bank.zero_grad()
    loss4b, pred4b, _ = tr.forward_backward(z, msg, eps, t, ctx)
    g4b = tr.bank.grad[:n]
    assert torch.equal(pred4b, pred4)                                             # the forward pass is bit-identical ...
    assert abs(loss4b.item() - loss4.item()) < 5e-6 * loss4.item()                # ... the MSE reduction uses fp32 atomics too (one run in ~20 exceeded 1e-6)
    spread_l2 = l2rel(g4b, g4)
    spread_max = ((g4b - g4).abs().max() / g4.abs().max()).item()
    print(f"atomic accumulation spread between two identical steps: l2rel {spread_l2:.2e}, max |diff| / max |g| {spread_max:.2e}")
    assert spread_l2 < 2e-5 and spread_max < 2e-5, (spread_l2, spread_max)        # fp32 reassociation only (eps = 6e-8 per add)
    tr.bank.zero_grad()
    acc = torch.zeros_like(g4)
    losses = []
    for i in range(B):
        cur["sl"] = slice(i, i + 1)
        tr.bank.zero_grad()
        li, pi, ci = tr.forward_backward(z[i:i + 1], msg[i:i + 1], eps[i:i + 1], t[i:i + 1], ctx[i:i + 1])
        acc += tr.bank.grad[:n] / B
        losses.append(li.item())
        # different tiles / kernels at the two batch sizes: bf16 rounding differences only
        ep, ec = l2rel(pred4[i:i + 1], pi), l2rel(clean4[i:i + 1], ci)
        print(f"sample {i}: pred l2rel {ep:.3e}, clean l2rel {ec:.3e}")
        assert ep < 3e-2 and ec < 3e-2, (i, ep, ec)   # measured 1.2-1.6e-2: the level of the HIP-vs-oracle difference itself
    torch.cuda.synchronize()
    assert torch.isfinite(g4).all()
    assert abs(loss4.item() - sum(losses) / B) < 2e-2 * loss4.item(), (loss4.item(), losses)
    e = l2rel(g4, acc)
    print(f"batch-4 twin step vs four batch-1 steps: loss {loss4.item():.5e} vs {sum(losses) / B:.5e}, flat gradient l2rel {e:.3e}")
    assert e < 5e-2, e


def test_full_size_batch8_rank320_twin_step_equals_mean_of_batch1_steps():
    """BASELINE config 3 per GPU (rank 320, batch 8) end to end at full size, through the data-parallel form of the step
    (single-rank RCCL group, AQL_FORCE_ALLREDUCE=1: wide-rank weight gradients held back, exchange buckets with one
    all-reduce each through aql_comm_*, early buckets forked from the backward hook): ONE batch-8 twin step equals the mean of eight batch-1 steps -- predicted noise per sample, loss, the whole
    flat gradient (LoRA + mapper).  The batch-1 step at rank 320 is pinned to the CPU oracle by
    test_full_size_ppft_gradients_vs_oracle[320]; this closes the shapes that only exist at batch 8 (tests/dp_config3_worker.py)."""
    import json, os, subprocess, sys
    from tests.conftest import ROOT
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29549")
    out = subprocess.run([sys.executable, "-m", "tests.dp_config3_worker"], env=env, cwd=ROOT, capture_output=True, text=True,
                         timeout=1500)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    print(rec)
    # (wide-rank problems: table "x" = 128 x 160 tiles for a side of 320, round 6; "w" = 128 x 128 tiles)
    assert (rec["overlap"] or rec["bucketed"]) and rec["twin"] and ("x" in rec["problem_kinds"] or "w" in rec["problem_kinds"]), rec
    ranges = rec["ranges"]
    # overlapped exchange: 3 early (up path, 245 MB) + 4 late buckets covering the mapper gradient too; fallback: 8 buckets
    end = rec["numel"] if rec["overlap"] else rec["n_lora"]
    assert len(ranges) >= 6 and ranges[0][0] == 0 and ranges[-1][1] == end, rec
    assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
    assert rec["grad_finite"] and rec["params_finite"] and rec["graph_buckets"] == len(ranges)
    assert not rec["overlap"] or (rec["n_graphs"] == 1 and any(a[1] == rec["n_early"] for a in ranges)), rec
    if rec["overlap"]:
        # three backward legs (up path | mid + down_blocks.3/.2 | down_blocks.1) end with a hook-driven exchange; what is left for
        # the end of backward at rank 320 (down_blocks.0 + the mapper) is ONE bucket of <= 64 MB (round-3 review: 4 x ~75 MB)
        cuts = rec["cuts"]
        assert rec["n_legs"] == 3 and all(any(a[1] == c for a in ranges) for c in cuts), (cuts, ranges)
        late = [a for a in ranges if a[0] >= cuts[-1]]
        assert len(late) == 1 and 4 * (late[0][1] - late[0][0]) <= 64 << 20, late
    # different tiles / kernels at the two batch sizes: bf16 rounding differences only (batch 4 / rank 32 measures 1.2-1.6e-2)
    assert max(rec["pred_l2rel"]) < 3e-2 and max(rec["clean_l2rel"]) < 3e-2, rec
    assert abs(rec["loss8"] - rec["loss_mean_b1"]) < 2e-2 * rec["loss8"], rec
    # measured 0.052 (LoRA part) / 0.027 (mapper): rank 320 is about twice as noisy against the oracle as rank 32 too (0.067 vs
    # 0.055 worst tensor above; the batch-4 / rank-32 twin test measures 0.014); bound = 1.5x measured
    assert rec["grad_l2rel"] < 8e-2 and rec["grad_lora_l2rel"] < 8e-2 and rec["grad_mapper_l2rel"] < 5e-2, rec
    # captured step == eager step from the same parameters, two optimizer steps deep (the first AdamW steps at lr 1e-4 move
    # 136 M parameters by +-1e-4 each: the loss itself jumps by an order of magnitude, identically in both forms)
    assert all(abs(a - b) < 2e-2 * abs(a) for a, b in zip(rec["eager_losses"], rec["graph_losses"])), rec
    assert rec["graph_vs_eager_param_relerr"] < 2e-3, rec


@pytest.mark.parametrize("rank,batch", [(32, 4), (320, 8)])
def test_default_captured_exchange_equals_unexchanged_step(rank, batch):
    """BASELINE configs 2 and 3 at full size: with nothing but a process group present (AQL_COMM unset) the trainer takes the
    captured, hook-driven aql_comm_* exchange (ONE step graph, three legs under backward; ppft_train.py:905-912,1058) -- and on a
    single-rank communicator, where the mean is the identity, that step equals the un-exchanged single-GPU step: loss,
    gradients and parameters up to the order of the fp32 atomics (measured: gradients 4e-8, parameters 9e-9 relative) (tests/dp_identity_worker.py)."""
    import json, os, subprocess, sys
    from tests.conftest import ROOT
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29551")
    for k in ("AQL_COMM", "AQL_FORCE_ALLREDUCE"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, "-m", "tests.dp_identity_worker", str(rank), str(batch)], env=env, cwd=ROOT,
                         capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    print(rec)
    assert rec["exchange_overlap"] and rec["exchange_n_graphs"] == 1 and rec["exchange_comm_note"] == "ok", rec
    assert not rec["plain_overlap"], rec
    rg = rec["exchange_ranges"]
    assert rg[0][0] == 0 and rg[-1][1] == rec["numel"] and all(a[1] == b[0] for a, b in zip(rg, rg[1:])), rg
    # (the loss itself is an fp32 atomic sum: equal to its last bit or two, measured 1.1e-7)
    assert rec["finite"] and abs(rec["plain_losses"][0] - rec["exchange_losses"][0]) <= 1e-6 * abs(rec["plain_losses"][0]), rec
    assert rec["grad_relerr"] < 1e-4 and rec["param_relerr"] < 2e-3, rec     # (AdamW's first steps are lr * sign(g): see the bucketed test)
    assert abs(rec["plain_losses"][1] - rec["exchange_losses"][1]) < 2e-2 * abs(rec["plain_losses"][1]), rec


def test_secret_decoder_vs_torchvision_live():
    """SURVEY.md section 8(c) option 2: when the box's own Python has torchvision, pin the decoder to the real
    ``efficientnet_b1`` (utils/models.py:84-96).  torchvision is absent from the build image, so this normally skips."""
    tv = pytest.importorskip("torchvision")
    from aqualora_amd import metrics
    from oracle.decoder_oracle import secret_decoder
    net = tv.models.efficientnet_b1(weights=None)
    net.classifier[1] = torch.nn.Linear(1280, 96)
    dec = _synthetic_decoder(48)
    net.load_state_dict({k[len("model."):]: v for k, v in dec.state_dict().items()})
    net.eval()
    x = T("dec.tv.x", (2, 3, 512, 512), 0.5).clamp(-1, 1)
    with torch.no_grad():
        want = net(x).view(-1, 48, 2)
        ours_cpu = secret_decoder({k: v.clone() for k, v in dec.state_dict().items()}, x, 48)
    got = dec.to(DEV).eval()(x.to(DEV))
    assert relerr(ours_cpu, want) < 1e-4 and relerr(got, want) < 1e-3
    assert torch.equal(metrics.extract_bits(got).cpu(), metrics.extract_bits(want))


def test_two_rank_rccl_replicas_stay_identical():
    """Two processes, two GPUs, RCCL: ranks start from DIFFERENT seeds (the reference default, seed=None) and different
    data; after the construction-time broadcast and 3 steps of averaged gradients the trainable state must be identical on
    both ranks.  Skips on 1-GPU boxes (the driver's multi-GPU node runs it)."""
    import json, os, subprocess, sys
    from tests.conftest import ROOT
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                          "127.0.0.1", "--master-port", "29547", "-m", "tests.dp_two_rank_worker"], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert res["world"] == 2 and res["rccl_ranks_seen"] == 2
    assert res["init_differs_before_broadcast"] and res["params_equal_after_init"]
    assert res["params_equal_after_steps"] and res["params_moved"], res
    # the validation the opt-in waits for (dp.make_comm): the captured aql_comm_* exchange against the torch.distributed one, from
    # the same initialisation and data -- exchanged gradients and parameters after three steps (fp32 atomics: last-bit noise)
    assert res["default_is_overlap"] is False
    if res["overlap_used"]:
        assert res["aql_comm_vs_torch_dist_grad_relerr"] < 1e-4 and res["aql_comm_vs_torch_dist_param_relerr"] < 2e-3, res


def test_reference_rounding_mode_matches_autocast_restatement_per_element():
    """AQL_REF_ROUNDING=1 inserts the reference's three bf16 roundings of the LoRA branch (utils/lora_modules.py:13-19 under
    autocast: bf16(down(x)), bf16(T*S), bf16(up(.)), then the bf16 add at :61).  With the same rounding points an fp32 torch
    restatement can be matched ELEMENT BY ELEMENT: every output within one bf16 ulp (2^-8 relative + accumulation-order noise),
    instead of the 1.5e-2-of-max bound of the fused single-accumulator kernel."""
    from aqualora_amd import lora as AL, ops
    r16 = lambda t: t.to(torch.bfloat16).float()   # noqa: E731
    old = ops.REF_ROUNDING
    ops.REF_ROUNDING = True
    try:
        for tag, cin, cout, n, r in (("rr_a", 320, 320, 256, 32), ("rr_b", 768, 640, 77, 32), ("rr_c", 1280, 320, 64, 8)):
            host = AL.LoRACompatibleLinear(cin, cout, device=DEV, dtype=torch.bfloat16)
            ll = AL.LoRALinearLayer(cin, cout, r, device=DEV, dtype=torch.float32)
            with torch.no_grad():
                host.weight.copy_(T(f"{tag}.w", (cout, cin), cin ** -0.5))
                host.bias.copy_(T(f"{tag}.b", (cout,), 0.02))
                ll.down.weight.copy_(T(f"{tag}.down", (r, cin), 1.0 / r))
                ll.up.weight.copy_(T(f"{tag}.up", (cout, r), 0.05))
            host.set_lora_layer(ll)
            x = T(f"{tag}.x", (2, n, cin), device=DEV).to(torch.bfloat16)
            S = (T(f"{tag}.S", (2, r), 0.3, DEV) + 1.0)
            y = AL.CustomLoRACompatibleLinearforward(host, x, S).float().cpu()
            xf = x.float().cpu().double()
            W, b = r16(host.weight.float().cpu()).double(), r16(host.bias.float().cpu()).double()
            A, Bu = r16(ll.down.weight.cpu()).double(), r16(ll.up.weight.cpu()).double()
            S16 = r16(S.cpu()).double()
            base = r16((xf @ W.t() + b).float()).double()
            Tm = r16((xf @ A.t()).float()).double()
            Ts = r16((Tm * S16[:, None, :]).float()).double()
            lo = r16((Ts @ Bu.t()).float()).double()
            ref = r16((base + lo).float())
            err = (y - ref).abs()
            # one bf16 ulp of the result plus one ulp of either summand (an fp32 accumulation-order flip of a rounding upstream)
            # ... and the propagation of a one-ulp flip of a single T (or Ts) element through the up projection
            flip = 2.0 ** -8 * 2.0 * float((Tm.abs().max() * S16.abs().max() * Bu.abs().max()).detach())
            bound = 2.0 ** -8 * (ref.abs() + base.abs().float() + lo.abs().float()) + flip + 1e-6
            frac_exact = (y == ref).float().mean().item()
            print(f"ref-rounding {tag}: {100 * frac_exact:.2f} % of elements bit-equal, worst excess over the bound "
                  f"{(err - bound).max().item():.2e}")
            assert (err <= bound).all() and frac_exact > 0.97, (tag, frac_exact)
    finally:
        ops.REF_ROUNDING = old


def test_tiny_ppft_step_vs_golden(golden):
    from aqualora_amd.ppft import PPFTTrainer
    from aqualora_amd.watermark import MapperNet, SecretEncoder
    g = golden("tiny_ppft.npz")
    unet, keys, lw = _gpu_tiny()
    inp = ppft_inputs(device=DEV)
    mapper = MapperNet(48, TINY_RANK)
    with torch.no_grad():
        mapper.bit_embeddings.weight.copy_(inp["E"])
    enc = SecretEncoder(48, base_res=8, resolution=16)
    tr = PPFTTrainer(unet, mapper, enc, TINY_RANK, learning_rate=1e-4)
    # inject the golden watermark residual (the fixture's wm is synthetic, not an encoder output)
    tr.sec_encoder.encode = lambda m, out_scale=1.0: inp["wm"]
    loss, pred, clean = tr.forward_backward(inp["z"], inp["msg"], inp["eps"], inp["t"], inp["ctx"])
    assert relerr(clean, g["clean"]) < 5e-2 and relerr(pred, g["pred"]) < 5e-2
    assert abs(loss.item() - float(g["loss"])) < 0.15 * float(g["loss"])
    # Gradients.  bf16 activation storage perturbs per-tensor LoRA gradients of this tiny config by 10-20 % against
    # the fp32 golden (the bf16-mirroring oracle shows the same, see DESIGN.md "tolerances"), so the tight check is
    # HIP vs the bf16-mirroring oracle and the golden check is a looser sanity bound.
    from oracle import ppft_oracle as O
    cpu = ppft_inputs()
    lo = {k: (d.clone().requires_grad_(True), u.clone().requires_grad_(True)) for k, (d, u) in lw.items()}
    Eo = cpu["E"].clone().requires_grad_(True)
    lo_loss, _, _, _ = O.ppft_loss(tiny_unet().state_dict(), TINY, lo, Eo, cpu["msg"], cpu["z"], cpu["wm"], cpu["eps"],
                                   cpu["t"], cpu["ctx"], bf16=True)
    lo_loss.backward()
    assert abs(loss.item() - lo_loss.item()) < 0.05 * lo_loss.item()

    def l2rel(a, b):
        a, b = a.detach().double().cpu().flatten(), torch.as_tensor(np.asarray(b)).double().flatten()
        return ((a - b).norm() / (b.norm() + 1e-30)).item()

    worst_o = worst_g = 0.0
    all_o = []
    gsq = 0.0
    for k in keys:
        lay = unet.get_submodule(k).lora_layer
        for got, want in ((lay.down.weight.grad, lo[k][0].grad), (lay.up.weight.grad, lo[k][1].grad)):
            gsq += got.double().pow(2).sum().item()
            if want.norm() > 0.05 * max(p.grad.norm() for pr in lo.values() for p in pr):
                all_o.append(l2rel(got, want.reshape(got.shape)))
                worst_o = max(worst_o, all_o[-1])
    for name in g.files:
        if name.startswith("g."):
            key, which = name[2:].rsplit(".", 1)
            lay = unet.get_submodule(key).lora_layer
            worst_g = max(worst_g, l2rel((lay.down if which == "down" else lay.up).weight.grad, g[name]))
    # measured on MI355X: median 0.05, worst 0.11 (uniform over site types: bf16 rounding of the backward signal)
    assert worst_o < 0.15 and float(np.median(all_o)) < 0.08, (worst_o, float(np.median(all_o)))
    assert worst_g < 0.15, worst_g
    assert l2rel(mapper.bit_embeddings.weight.grad, Eo.grad) < 0.10
    assert abs(gsq ** 0.5 - float(g["total_norm"])) < 0.1 * float(g["total_norm"])
    tr.exchange_gradients()
    tr.optimizer_step()
    k0 = keys[0]
    lay = unet.get_submodule(k0).lora_layer
    assert relerr(lay.down.weight, g["p." + k0 + ".down"]) < 1e-3
    assert relerr(lay.up.weight, g["p." + k0 + ".up"]) < 5e-2
    assert relerr(mapper.bit_embeddings.weight, g["p.mapper"]) < 1e-3
    total = float(g["total_norm"])
    assert abs(tr.grad_norm() - total) < 0.1 * total


def test_fused_step_prologue_equals_generic_head():
    """aql_ppft_prologue (mapper, noisy twin latents at conv_in's packed width, text states twice, timestep embedding, [0 | S] scale
    rows, zeroed dS accumulator in ONE launch) against the generic 15-launch head of the twin step: identical prediction, clean
    target, loss and S bits; gradients equal up to the fp32-atomic order of the weight-gradient kernels."""
    import aqualora_amd.ppft as P
    from aqualora_amd.watermark import MapperNet, SecretEncoder
    outs = []
    for fused in (True, False):
        unet, keys, lw = _gpu_tiny()
        inp = ppft_inputs(device=DEV)
        mapper = MapperNet(48, TINY_RANK)
        with torch.no_grad():
            mapper.bit_embeddings.weight.copy_(inp["E"])
        tr = P.PPFTTrainer(unet, mapper, SecretEncoder(48, base_res=8, resolution=16), TINY_RANK, learning_rate=1e-4)
        tr.sec_encoder.encode = lambda m, out_scale=1.0: inp["wm"]
        old = P._PROLOGUE
        P._PROLOGUE = fused
        try:
            used = []
            orig = tr._twin_prologue
            tr._twin_prologue = lambda *a: (used.append(1), orig(*a))[1]
            loss, pred, clean = tr.forward_backward(inp["z"], inp["msg"], inp["eps"], inp["t"], inp["ctx"])
        finally:
            P._PROLOGUE = old
        assert bool(used) == fused
        grads = torch.cat([unet.get_submodule(k).lora_layer.down.weight.grad.flatten() for k in keys] +
                          [mapper.bit_embeddings.weight.grad.flatten()])
        outs.append((loss.item(), pred.float().clone(), clean.float().clone(), grads.clone()))
    (l0, p0, c0, g0), (l1, p1, c1, g1) = outs
    assert torch.equal(p0, p1) and torch.equal(c0, c1), ((p0 - p1).abs().max().item(), (c0 - c1).abs().max().item())
    assert abs(l0 - l1) <= 1e-6 * abs(l1)   # the MSE kernel adds its block sums with fp32 atomics: last-bit order dependence
    assert float((g0 - g1).norm() / g1.norm()) < 1e-5


def test_checkpoint_roundtrip_and_consumer_contract(tmp_path, golden):
    from aqualora_amd.checkpoint import load_lora_state, save_lora_weights
    from aqualora_amd.lora import inject_lora
    from aqualora_amd.watermark import MapperNet
    from safetensors.torch import load_file
    unet, keys, lw = _gpu_tiny()
    mp = MapperNet(48, TINY_RANK)
    save_lora_weights(str(tmp_path), unet, mp)
    sd = load_file(str(tmp_path / "pytorch_lora_weights.safetensors"))
    assert sorted(sd.keys()) == list(golden("checkpoint_layout.npz")["names"])
    # create_wm_lora.py:24-41 branches on these substrings
    for k in sd:
        assert "unet" in k and (("attn" in k or "ff" in k) or ("proj_in" in k or "proj_out" in k))
        assert ("up.weight" in k) or ("down.weight" in k)
        assert ".alpha" not in k
    back = load_lora_state(str(tmp_path))
    for k in keys:
        assert torch.equal(back[k + ".down.weight"], lw[k][0]) and torch.equal(back[k + ".up.weight"], lw[k][1])
    assert list(torch.load(str(tmp_path / "mapper.pt")).keys()) == ["bit_embeddings.weight"]


def test_jpeg_layer_vs_reference_golden(golden):
    from aqualora_amd.noise import JpegCompression
    g = golden("jpeg.npz")
    layer = JpegCompression()
    for tag, shape in (("a", (2, 3, 64, 64)), ("b", (1, 3, 50, 44))):
        x = T(f"jpeg.{tag}.x", shape, 0.5, DEV).requires_grad_(True)
        y = layer([x, None])[0]
        y.backward(T(f"jpeg.{tag}.dy", shape, device=DEV))
        assert relerr(y, g[f"{tag}.y"]) < 1e-5 and relerr(x.grad, g[f"{tag}.dx"]) < 1e-5
    # full-size property checks (512x512): idempotence of the projection and linearity
    x = T("jpeg.big", (2, 3, 512, 512), 0.5, DEV)
    y = layer([x.clone(), None])[0]
    yy = layer([y.clone(), None])[0]
    assert relerr(layer([2.0 * x, None])[0], 2.0 * y) < 1e-5
    assert y.shape == x.shape and torch.isfinite(yy).all()


def _synthetic_decoder(bits=48):
    from aqualora_amd import synth
    from aqualora_amd.decoder import SecretDecoder
    dec = SecretDecoder(bits)
    with torch.no_grad():
        for name, t in dec.state_dict().items():
            if name.endswith("num_batches_tracked"):
                continue
            if name.endswith("running_var"):
                t.copy_(synth.normal(name, t.shape, 0.2, SEED).abs() + 0.5)
            elif name.endswith("running_mean"):
                t.copy_(synth.normal(name, t.shape, 0.1, SEED))
            elif t.dim() >= 2:
                t.copy_(synth.normal(name, t.shape, (1.0 / t[0].numel()) ** 0.5, SEED))
            elif name.endswith(".1.weight"):  # BN gamma; the residual-branch (project) BN is damped so that the
                last = (".block.3.1." in name) or (".block.2.1." in name and "features.1." in name)  # net stays O(1)
                t.copy_((1.0 + synth.normal(name, t.shape, 0.1, SEED)) * (0.3 if last else 1.0))
            else:
                t.copy_(synth.normal(name, t.shape, 0.05, SEED))
    return dec


def test_secret_decoder_vs_oracle_bits_exact():
    """fp32 logits within 1e-3 of the CPU oracle and the extracted bits identical (north star: bits bit-exact)."""
    from aqualora_amd import metrics
    from oracle.decoder_oracle import secret_decoder
    dec = _synthetic_decoder(48)
    sd = {k: v.clone() for k, v in dec.state_dict().items()}
    dec = dec.to(DEV).eval()
    for shape in ((2, 3, 512, 512), (1, 3, 576, 640), (3, 3, 96, 80)):
        x = T("dec.x" + str(shape), shape, 0.5).clamp(-1, 1)
        with torch.no_grad():
            want = secret_decoder(sd, x, 48)
        got = dec(x.to(DEV))
        assert got.shape == want.shape == (shape[0], 48, 2)
        assert relerr(got, want) < 1e-3, relerr(got, want)
        # north star: extracted bits bit-exact -- EVERY bit, no margin mask (fp32 kernels against the fp32 oracle)
        bits_got, bits_want = metrics.extract_bits(got).cpu(), metrics.extract_bits(want)
        mism = int((bits_got != bits_want).sum())
        margin = (want[..., 0] - want[..., 1]).abs()
        print(f"decoder {shape}: {mism} of {bits_want.numel()} bits differ; smallest logit margin "
              f"{margin.min().item():.3e} (logit scale {want.abs().max().item():.3e}, logits relerr {relerr(got, want):.2e})")
        assert mism == 0, (shape, mism)
    msg = metrics.extract_bits(want)
    acc, tpr = metrics.tpr_at_fpr(metrics.extract_bits(got).cpu(), msg, 1e-6)
    assert acc == 1.0 and tpr == 1.0


def test_secret_decoder_training_step_vs_oracle():
    """train()-mode EfficientNet-B1 (BatchNorm batch statistics, stochastic depth, dropout) forward, BCE loss and the
    full backward (all 301 parameter tensors + the input image) against torch autograd on the CPU restatement, with the
    random masks pinned; running statistics must move identically."""
    decoder_training_step_parity(2, 96, 80)


def decoder_training_step_parity(B, H, W, grad_tol=5e-2, image_grad_tol=2e-2):
    from aqualora_amd import decoder as D
    from oracle.decoder_oracle import secret_decoder_train
    torch.manual_seed(0)
    bits = 48
    dec = _synthetic_decoder(bits)
    sd = {k[len("model."):]: v.clone().float() for k, v in dec.state_dict().items()}
    params = {k: v.requires_grad_(True) for k, v in sd.items() if "running" not in k and "num_batches" not in k}
    nblk = 23
    sd_noise = [torch.bernoulli(torch.full((B,), 1.0 - 0.2 * i / nblk)) / (1.0 - 0.2 * i / nblk) for i in range(nblk)]
    drop = torch.bernoulli(torch.full((B, 1280), 0.8)) / 0.8
    x = T("dect.x", (B, 3, H, W), 0.5).clamp(-1, 1)
    msg = (T("dect.m", (B, bits), 1.0) > 0).long()
    target = torch.nn.functional.one_hot(msg, 2).float()

    xr = x.clone().requires_grad_(True)
    want = secret_decoder_train({**sd, **params}, xr, bits, sd_noise, drop)
    loss_r = torch.nn.functional.binary_cross_entropy_with_logits(want, target)
    loss_r.backward()

    dec = dec.to(DEV).train()
    xg = x.to(DEV).requires_grad_(True)
    got = dec(xg, sd_noise=sd_noise, drop_mask=drop)
    loss_g = D.bce_with_logits(got, target.to(DEV))
    loss_g.backward()

    assert relerr(got, want) < 2e-3, relerr(got, want)
    assert abs(loss_g.item() - loss_r.item()) < 1e-4 * max(1.0, abs(loss_r.item()))
    assert l2rel(xg.grad, xr.grad) < image_grad_tol, l2rel(xg.grad, xr.grad)
    worst, n = 0.0, 0
    for name, p in dec.model.named_parameters():
        ref = params[name].grad
        assert p.grad is not None and p.grad.shape == ref.shape, name
        if ref.norm() > 1e-6 * (1 + params[name].detach().norm()):
            e = l2rel(p.grad, ref)
            worst = max(worst, e)
            n += 1
    assert n > 250 and worst < grad_tol, (n, worst)
    for name, b in dec.model.named_buffers():
        if "running" in name:
            assert relerr(b, sd[name]) < 1e-3, name
    dec.eval()                       # folded inference weights are rebuilt from the moved running statistics
    assert dec(x.to(DEV)).shape == (B, bits, 2)
    return dict(logits=relerr(got, want), image_grad=l2rel(xg.grad, xr.grad), worst_param_grad=worst, n_grads=n)


def test_decoder_bn_stochastic_depth_skip_fusion_equals_the_three_launch_sets():
    """Round 6, the first fusion of the rob-finetune decoder step: the last BatchNorm of an MBConv block with its stochastic-depth scale
    and skip connection in ONE apply pass (aql_bn_train_fwd_res / aql_bn_train_bwd_rs) against BatchNorm -> chan-scale -> add: logits,
    running statistics, every parameter gradient and the image gradient within the run-to-run spread of the step's own fp32 atomics
    (the fused pass rounds where the launches rounded)."""
    from aqualora_amd import decoder as D
    torch.manual_seed(0)
    B, bits = 4, 48
    dec = _synthetic_decoder(bits).to(DEV).train()
    state = {k: v.clone() for k, v in dec.state_dict().items()}
    nblk = 23
    sd_noise = [torch.bernoulli(torch.full((B,), 1.0 - 0.2 * i / nblk)) / (1.0 - 0.2 * i / nblk) for i in range(nblk)]
    drop = torch.bernoulli(torch.full((B, 1280), 0.8)) / 0.8
    x = T("decf.x", (B, 3, 160, 128), 0.5).clamp(-1, 1).to(DEV)
    target = torch.nn.functional.one_hot((T("decf.m", (B, bits), 1.0) > 0).long(), 2).float().to(DEV)

    def run(fused):
        D.FUSE_BN_RES = fused
        dec.load_state_dict(state)
        for p in dec.parameters():
            p.grad = None
        xg = x.clone().requires_grad_(True)
        got = dec(xg, sd_noise=sd_noise, drop_mask=drop)
        D.bce_with_logits(got, target).backward()
        torch.cuda.synchronize()
        return got.detach().clone(), xg.grad.clone(), torch.cat([p.grad.reshape(-1) for p in dec.parameters()]), \
            torch.cat([b.float().reshape(-1) for n, b in dec.named_buffers() if "running" in n])
    try:
        a, a2, f = run(False), run(False), run(True)
    finally:
        D.FUSE_BN_RES = True
    # (the step's reductions use fp32 atomics: two runs of the SAME form differ in the last bits -- that spread is the yardstick)
    s_fwd = max(relerr(a2[0], a[0]), relerr(a2[3], a[3]))
    assert relerr(f[0], a[0]) <= 3 * s_fwd + 1e-6 and relerr(f[3], a[3]) <= 3 * s_fwd + 1e-6, (s_fwd, relerr(f[0], a[0]), relerr(f[3], a[3]))
    spread = max(l2rel(a2[1], a[1]), l2rel(a2[2], a[2]))
    assert l2rel(f[1], a[1]) <= 3 * spread + 1e-6 and l2rel(f[2], a[2]) <= 3 * spread + 1e-6, (spread, l2rel(f[1], a[1]), l2rel(f[2], a[2]))


def test_distortion_maps_vs_torch():
    """crop+bilinear resize, Gaussian blur (reflect), additive noise: forward and adjoint vs plain torch ops."""
    import torch.nn.functional as F
    from aqualora_amd import noise as NZ
    x = T("dist.x", (2, 3, 96, 80), 0.5, DEV).requires_grad_(True)
    y = NZ.crop_resize(x, 7, 5, 61, 50, 128, 96)
    xr = x.detach().clone().requires_grad_(True)
    yr = F.interpolate(xr[:, :, 7:68, 5:55], size=(128, 96), mode="bilinear", align_corners=False)
    dy = T("dist.dy", (2, 3, 128, 96), device=DEV)
    y.backward(dy); yr.backward(dy)
    assert relerr(y, yr) < 1e-5 and relerr(x.grad, xr.grad) < 1e-5
    for k, sigma in ((3, 0.7), (5, 4.0), (9, 2.0)):
        x2 = T("dist.x2", (2, 3, 64, 48), 0.5, DEV).requires_grad_(True)
        y2 = NZ.gaussian_blur(x2, k, sigma)
        taps = NZ.gaussian_taps(k, sigma, DEV)
        w2 = (taps[:, None] * taps[None, :])[None, None].repeat(3, 1, 1, 1)
        xr2 = x2.detach().clone().requires_grad_(True)
        yr2 = F.conv2d(F.pad(xr2, (k // 2,) * 4, mode="reflect"), w2, groups=3)
        dy2 = T("dist.dy2", (2, 3, 64, 48), device=DEV)
        y2.backward(dy2); yr2.backward(dy2)
        assert relerr(y2, yr2) < 1e-5 and relerr(x2.grad, xr2.grad) < 1e-5
    # kornia RandomGaussianBlur((3, 9), ...) as noises.py:68 calls it: 3 rows x 9 columns, one sigma per sample
    sig = torch.tensor([0.6, 3.5])
    x3 = T("dist.x3", (2, 3, 40, 56), 0.5, DEV).requires_grad_(True)
    y3 = NZ.gaussian_blur(x3, (3, 9), sig)
    xr3 = x3.detach().clone().requires_grad_(True)
    rows = []
    for i in range(2):
        ty, tx = NZ.gaussian_taps(3, float(sig[i]), DEV), NZ.gaussian_taps(9, float(sig[i]), DEV)
        w3 = (ty[:, None] * tx[None, :])[None, None].repeat(3, 1, 1, 1)
        rows.append(F.conv2d(F.pad(xr3[i:i + 1], (4, 4, 1, 1), mode="reflect"), w3, groups=3))
    yr3 = torch.cat(rows)
    dy3 = T("dist.dy3", (2, 3, 40, 56), device=DEV)
    y3.backward(dy3); yr3.backward(dy3)
    assert relerr(y3, yr3) < 1e-5 and relerr(x3.grad, xr3.grad) < 1e-5
    n = T("dist.n", (2, 3, 64, 48), device=DEV)
    z = NZ.add_gaussian_noise(x2.detach(), 0.1, clamp01=True, noise=n)
    assert torch.allclose(z, (x2.detach() + 0.1 * n).clamp(0, 1), atol=1e-6)  # kernel contracts to one fma
    # the additive noise is differentiable in x (kornia RandomGaussianNoise): identity gradient, masked by the clamp
    for clamp in (False, True):
        xn = (T("dist.xn", (2, 3, 64, 48), 0.4, DEV) + 0.5).requires_grad_(True)
        yn = NZ.add_gaussian_noise(xn, 0.3, clamp01=clamp, noise=n)
        assert yn.grad_fn is not None
        yn.backward(dy2)
        ref = (xn.detach() + 0.3 * n)
        mask = ((ref > 0) & (ref < 1)).float() if clamp else torch.ones_like(ref)
        assert torch.equal(xn.grad, dy2 * mask)
    out = NZ.distorsion_unit(T("dist.img", (1, 3, 512, 512), 0.2, DEV) + 0.5, "crop")
    assert out.shape == (1, 3, 512, 512)


def test_kornia_style_distortions_vs_oracle():
    """Colour jiggle (all 4 ops, two orders), centre rotation and sharpness: forward and adjoint vs the torch CPU
    restatement of kornia 0.6.12 (oracle/distort_oracle.py, parity unpinned: kornia is absent from the image)."""
    from aqualora_amd import noise as NZ
    from oracle import distort_oracle as DO

    def frac_bad(a, b, tol):
        a, b = a.float().cpu(), b.float().cpu()
        return ((a - b).abs() > tol * (1.0 + b.abs())).float().mean().item()

    x0 = (T("kd.x", (2, 3, 40, 56), 0.35, "cpu") + 0.5).clamp(0, 1)
    dy = T("kd.dy", (2, 3, 40, 56), device="cpu")
    for order in ((0, 1, 2, 3), (3, 1, 0, 2)):
        br, ct, sa, hu = [1.2, 0.75], [0.8, 1.25], [1.25, 0.8], [0.2, -0.15]
        xr = x0.clone().requires_grad_(True)
        yr = DO.color_jiggle(xr, br, ct, sa, hu, order)
        yr.backward(dy)
        xg = x0.to(DEV).requires_grad_(True)
        yg = NZ.color_jiggle(xg, br, ct, sa, hu, order)
        yg.backward(dy.to(DEV))
        assert frac_bad(yg, yr, 2e-5) == 0.0
        assert frac_bad(xg.grad, xr.grad, 1e-3) < 2e-3   # kinks (clamps, hue sectors, max/min ties) are measure-zero
    for ang in ([15.0, -160.0], [90.0, 0.0]):
        xr = x0.clone().requires_grad_(True)
        yr = DO.rotate(xr, ang)
        yr.backward(dy)
        xg = x0.to(DEV).requires_grad_(True)
        yg = NZ.rotate(xg, ang)
        yg.backward(dy.to(DEV))
        assert frac_bad(yg, yr, 1e-4) < 1e-3 and frac_bad(xg.grad, xr.grad, 1e-4) < 1e-3  # floor() ties at exact angles
    for fac in ([0.5, 10.0], [0.0, 1.0], [3.0, 0.25]):
        xr = x0.clone().requires_grad_(True)
        yr = DO.sharpness(xr, fac)
        yr.backward(dy)
        xg = x0.to(DEV).requires_grad_(True)
        yg = NZ.sharpness(xg, fac)
        yg.backward(dy.to(DEV))
        assert frac_bad(yg, yr, 1e-5) == 0.0 and frac_bad(xg.grad, xr.grad, 1e-4) < 1e-3
    img = (T("kd.img", (1, 3, 512, 512), 0.2, DEV) + 0.5).clamp(0, 1)
    for kind in ("color_jitter", "crop", "blur", "noise", "rotation", "sharpness"):
        out = NZ.eval_distorsion_unit(img, kind)
        assert out.shape == img.shape and torch.isfinite(out).all()
    assert NZ.distorsion_unit(img, "color_jitter").shape == img.shape
    nz = NZ.Noiser(["Identity", "Jpeg", "CropandResize", "GaussianBlur", "GaussianNoise", "ColorJitter"], [0, 0, 0, 0, 0, 1.0])
    out = nz([img * 2 - 1, None])
    assert out[0].shape == img.shape and out[1] is None


def test_stage1_pieces_vs_reference_golden_and_oracle(golden):
    """PRVL_loss (forward + gradient) against the reference's own outputs; SecretEncoder backward, gen_combined_latents
    and one full stage-1 step (encoder -> stand-in VAE decoder -> distortion -> train-mode decoder -> BCE + PRVL) against
    torch autograd on the CPU restatements."""
    import torch.nn.functional as F
    from aqualora_amd import decoder as D, noise as NZ, stage1 as S1
    from aqualora_amd.watermark import SecretEncoder
    from oracle import ppft_oracle as O, stage1_oracle as SO
    from oracle.decoder_oracle import secret_decoder_train
    from tests.common import prvl_case
    g = golden("stage1_prvl.npz")
    for i in (1, 2):
        a = torch.tensor(g[f"a{i}"]).to(DEV)
        b = torch.tensor(g[f"b{i}"]).to(DEV).requires_grad_(True)
        loss = S1.PRVL_loss(a, b)
        loss.backward()
        assert abs(loss.item() - float(g[f"loss{i}"])) < 2e-6 * max(1.0, float(g[f"loss{i}"]))
        assert np.allclose(b.grad.cpu().numpy(), g[f"grad_b{i}"], atol=1e-7)
    a, b = prvl_case(0, 1, 512, 512, 0.05)
    b = b.to(DEV).requires_grad_(True)
    loss = S1.PRVL_loss(a.to(DEV), b)
    loss.backward()
    assert abs(loss.item() - float(g["loss0"])) < 2e-6
    nz = b.grad.nonzero()
    assert len(nz) == int(g["grad_nnz0"]) and abs(b.grad.abs().sum().item() - float(g["grad_abs_sum0"])) < 1e-5
    assert [int(nz[:, 2].min()), int(nz[:, 2].max()), int(nz[:, 3].min()), int(nz[:, 3].max())] == g["grad_bbox0"].tolist()

    # SecretEncoder: trainable forward + backward
    B, bits = 3, 48
    enc = SecretEncoder(bits)
    with torch.no_grad():
        enc.secret_scaler[0].weight.copy_(T("s1.lw", (1024, bits), 0.2))
        enc.secret_scaler[0].bias.copy_(T("s1.lb", (1024,), 0.2))
        enc.secret_scaler[5].weight.copy_(T("s1.cw", (4, 4, 3, 3), 0.1))
        enc.secret_scaler[5].bias.copy_(T("s1.cb", (4,), 0.1))
    ref = [p.detach().clone().requires_grad_(True) for p in (enc.secret_scaler[0].weight, enc.secret_scaler[0].bias,
                                                              enc.secret_scaler[5].weight, enc.secret_scaler[5].bias)]
    msg = (T("s1.m", (B, bits)) > 0).float()
    lat = T("s1.lat", (B, 4, 64, 64))
    dy = T("s1.dy", (B, 4, 64, 64))
    cm_r = O.secret_encoder(msg, *ref)
    ((lat + cm_r) * dy).sum().backward()
    enc = enc.to(DEV)
    xo, cm = enc(lat.to(DEV), msg.to(DEV))
    (xo * dy.to(DEV)).sum().backward()
    assert relerr(cm, cm_r) < 1e-5
    for p, r in zip((enc.secret_scaler[0].weight, enc.secret_scaler[0].bias, enc.secret_scaler[5].weight,
                     enc.secret_scaler[5].bias), ref):
        assert l2rel(p.grad, r.grad) < 1e-4, l2rel(p.grad, r.grad)

    # gen_combined_latents, both branches
    wm = T("s1.wm", (B, 4, 64, 64), 0.1)
    for corner, sc in ((False, (1.0, 1.0)), (True, (1.37, 1.81))):
        wr = wm.clone().requires_grad_(True)
        out_r = SO.gen_combined_latents(lat, wr, 0.03, corner, sc)
        (out_r * dy).sum().backward()
        wg = wm.to(DEV).requires_grad_(True)
        out_g = S1.gen_combined_latents(lat.to(DEV), wg, 0.03, corner, sc)
        (out_g * dy.to(DEV)).sum().backward()
        assert relerr(out_g, out_r) < 1e-5 and l2rel(wg.grad, wr.grad) < 1e-5

    # one full step (post-warm-up schedule, identity distortion) with a stand-in linear "VAE decoder"
    B = 2
    dec = _synthetic_decoder(bits)
    sd = {k[len("model."):]: v.clone().float() for k, v in dec.state_dict().items()}
    dparams = {k: v.requires_grad_(True) for k, v in sd.items() if "running" not in k and "num_batches" not in k}
    Wd = T("s1.vae", (3, 4), 0.5)

    def vae_decode(w):
        def f(z):  # nearest x4 upsample + 1x1 mix, tanh-free so it is exactly linear: [B,4,64,64] -> [B,3,256,256]
            return torch.einsum("oc,bchw->bohw", w.to(z.device), F.interpolate(z, scale_factor=4.0, mode="nearest"))
        return f

    sd_noise = [torch.ones(B) for _ in range(23)]
    drop = torch.ones(B, 1280)
    msg = (T("s1.m2", (B, bits)) > 0).long()
    lat = T("s1.lat2", (B, 4, 64, 64))
    cm_r = O.secret_encoder(msg.float(), *ref)
    for r in ref:
        r.grad = None
    wml_r = SO.gen_combined_latents(lat, cm_r, 1.0, False)
    clean_r, wimg_r = vae_decode(Wd)(lat).detach(), vae_decode(Wd)(wml_r)
    prvl_r = SO.prvl_loss(clean_r, wimg_r)
    logits_r = secret_decoder_train({**sd, **dparams}, wimg_r, bits, sd_noise, drop)
    msgloss_r = F.binary_cross_entropy_with_logits(logits_r, F.one_hot(msg, 2).float())
    loss_r = SO.stage1_loss(msgloss_r, torch.zeros(()), prvl_r, False, 11, False)
    loss_r.backward()

    dec = dec.to(DEV).train()
    orig_forward = dec.forward
    dec.forward = lambda x: orig_forward(x, sd_noise=sd_noise, drop_mask=drop)
    for p in enc.parameters():
        p.grad = None
    step = S1.Stage1Step(enc, dec, vae_decode(Wd), NZ.Noiser(["Identity", "Jpeg"], [1.0, 0.0]))
    step.warmup = False
    out = step.losses(lat.to(DEV), msg.to(DEV), epochs_done=11, combine={"cornerfy_aug": False},
                      noiser_choice=[1.0, 0.0])
    out["loss"].backward()
    assert abs(out["prvl_loss"].item() - prvl_r.item()) < 1e-5 * max(1.0, prvl_r.item())
    assert abs(out["msgloss"].item() - msgloss_r.item()) < 2e-4 * max(1.0, msgloss_r.item())
    assert abs(out["loss"].item() - loss_r.item()) < 2e-4 * max(1.0, loss_r.item())
    for p, r in zip((enc.secret_scaler[0].weight, enc.secret_scaler[0].bias, enc.secret_scaler[5].weight,
                     enc.secret_scaler[5].bias), ref):
        assert l2rel(p.grad, r.grad) < 3e-2, l2rel(p.grad, r.grad)
    worst = max(l2rel(p.grad, dparams[n].grad) for n, p in dec.model.named_parameters()
                if dparams[n].grad.norm() > 1e-6 * (1 + dparams[n].detach().norm()))
    assert worst < 5e-2, worst

    # rob-finetune step: decoder-only update on distorted [0,1] images
    opt = torch.optim.AdamW(dec.parameters(), lr=1e-4)
    before = dec.model.classifier[1].weight.detach().clone()
    imgs = (T("s1.img", (B, 3, 128, 160), 0.2, DEV) + 0.5).clamp(0, 1)
    loss, acc = S1.rob_finetune_step(dec, opt, imgs, msg.to(DEV), distort=NZ.RobNoiser([0.0, 1.0, 0.0, 0.0, 0.0]))
    assert torch.isfinite(loss) and 0.0 <= acc.item() <= 1.0
    assert not torch.equal(before, dec.model.classifier[1].weight.detach())


def test_captured_step_equals_eager_step():
    """HIP-graph replay (two graphs + exchange) must train exactly like the eager step: same loss trajectory and the
    same parameters after 3 steps (up to fp32 atomic-order noise in the weight gradients)."""
    from aqualora_amd.ppft import PPFTTrainer
    from aqualora_amd.watermark import MapperNet, SecretEncoder
    inp = ppft_inputs(device=DEV)
    batch = dict(z=inp["z"], msg=inp["msg"], eps=inp["eps"], t=inp["t"], ctx=inp["ctx"].to(torch.bfloat16))
    results = []
    for mode in ("eager", "graph"):
        unet, keys, lw = _gpu_tiny()
        mapper = MapperNet(48, TINY_RANK)
        with torch.no_grad():
            mapper.bit_embeddings.weight.copy_(inp["E"])
        enc = SecretEncoder(48, base_res=8, resolution=16)
        with torch.no_grad():
            enc.secret_scaler[0].weight.copy_(T("cap.lin.w", (64, 48), 48 ** -0.5))
            enc.secret_scaler[0].bias.copy_(T("cap.lin.b", (64,), 0.1))
            enc.secret_scaler[5].weight.copy_(T("enc.conv.w", (4, 4, 3, 3), 0.05))
        tr = PPFTTrainer(unet, mapper, enc, TINY_RANK, learning_rate=1e-3)
        run = tr.step if mode == "eager" else tr.capture(batch, warmup=0)
        losses = [float(run(**batch)) for _ in range(3)]
        torch.cuda.synchronize()
        results.append((losses, tr.bank.flat.clone()))
    (le, pe), (lg, pg) = results
    assert all(abs(a - b) < 2e-3 * abs(a) for a, b in zip(le, lg)), (le, lg)
    assert relerr(pg, pe) < 2e-3
    assert le[-1] < le[0]  # and it actually trains


def test_create_watermark_lora_vs_reference_golden(golden):
    from aqualora_amd.inference import create_watermark_lora
    from aqualora_amd.watermark import MapperNet
    g = golden("create_wm_lora.npz")
    r = 320
    sd = {}
    for k in g.files:
        if k == "msg":
            continue
        base, which = k.rsplit(".", 2)[0], k.rsplit(".", 2)[1]
        shape = g[k].shape
        sd[k] = T(base + "." + which, shape, 1.0 / r if which == "down" else 0.05)
    mp = MapperNet(48, r).to(DEV)
    with torch.no_grad():
        mp.bit_embeddings.weight.copy_(T("cwl.E", (48, r)))
    hid, out = create_watermark_lora(sd, mp, str(g["msg"]), 1.03)
    assert hid == str(g["msg"]) and sorted(out) == sorted(k for k in g.files if k != "msg")
    for k in out:
        assert relerr(out[k], g[k]) < 1e-6, k


def test_fuse_lora_and_ddim_sampler():
    from aqualora_amd.checkpoint import lora_state_dict
    from aqualora_amd.inference import ddim_sample, ddim_timesteps, fuse_lora
    from oracle import ppft_oracle as O
    assert ddim_timesteps(50)[:3] == [981, 961, 941] and ddim_timesteps(50)[-1] == 1
    unet, keys, lw = _gpu_tiny()
    inp = ppft_inputs(device=DEV)
    x = inp["z"]
    with torch.no_grad():
        y_lora = unet(x, inp["t"], inp["ctx"], cross_attention_kwargs={"scale": 1.0}).sample.float()
    sd = lora_state_dict(unet, keys)
    fuse_lora(unet, sd, 1.0, keys)
    assert all(unet.get_submodule(k).lora_layer is None for k in keys)
    with torch.no_grad():
        y_fused = unet(x, inp["t"], inp["ctx"]).sample.float()
    assert relerr(y_fused, y_lora) < 3e-2
    # sampler: graph replay == eager loop == oracle update rule driven by the same U-Net
    ctx_u = torch.zeros_like(inp["ctx"])
    lat = T("ddim.lat", (2, 4, 16, 16), device=DEV)
    a = ddim_sample(unet, inp["ctx"], ctx_u, lat, num_inference_steps=5, guidance_scale=7.5, graph=True)
    b = ddim_sample(unet, inp["ctx"], ctx_u, lat, num_inference_steps=5, guidance_scale=7.5, graph=False)
    assert torch.equal(a, b)
    xr = lat.clone()
    ts = ddim_timesteps(5)
    ctx2 = torch.cat([ctx_u, inp["ctx"]])
    for t in ts:
        with torch.no_grad():
            e = unet(torch.cat([xr, xr]), torch.full((4,), t, device=DEV), ctx2).sample.float().contiguous()
        xr = O.ddim_step(xr.cpu(), e[:2].cpu(), e[2:].cpu(), t, t - 200, 7.5).to(DEV)
    assert relerr(a, xr) < 1e-4


def test_weight_side_lora_form_equals_the_activation_side_branch():
    """The opt-in weight-side form of the LoRA linear (AQL_WSIDE=1: per-sample effective weights W + Bup.diag(S_b).A through
    aql_gemm_bf16_sw, per-sample dY^T X, aql_wside_reduce) against the default activation-side branch on
    the same inputs: outputs, dX, dS, dA, dBup of a square 320 -> 320 site and of the feed-forward pair (GEGLU forward and backward
    epilogues with per-sample weights) at rank 320, 1024 and 4096 tokens per sample: relative L2 < 3e-2 (measured 2-5e-3: the bf16
    rounding of the effective weight instead of the bf16 rounding of T).  tools/probe_wside.py prints the numbers."""
    import os, subprocess, sys
    from tests.conftest import ROOT
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "probe_wside.py")], cwd=ROOT, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0 and "ALL PASS" in out.stdout and "FAIL" not in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_bench_under_torchrun_with_rccl_collective():
    """The N>1 launch path on one GPU: bench.py under torch.distributed.run with the gradient all-reduce forced through
    RCCL (AQL_FORCE_ALLREDUCE=1).  Regression test: RCCL's watchdog thread calls hipEventQuery while the step is being
    captured into HIP graphs, which aborts the process unless the capture is thread-local."""
    import json, os, subprocess, sys
    from tests.conftest import ROOT
    env = dict(os.environ, AQL_FORCE_ALLREDUCE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 1 and rec["value"] > 0 and rec["config"]["hip_graph"] is True
    assert "roofline" in rec and rec["roofline"]["frac"] > 0


def test_bucketed_exchange_equals_single_flush():
    """Data-parallel forms of the step on one GPU (single-rank RCCL group, collectives forced), at rank 8 (grouped problems)
    and rank 40 (wide problems): (a) the overlapped exchange through aql_comm_* -- early buckets forked from the backward hook,
    late buckets behind the last weight-gradient launch, eager and captured into ONE graph; (b) the torch.distributed
    fallback -- bucket graphs with an async all-reduce between them.  Parameters after 3 steps equal the single-GPU form's."""
    import json, os, subprocess, sys
    from tests.conftest import ROOT
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29547")
    env.pop("AQL_FORCE_ALLREDUCE", None)
    out = subprocess.run([sys.executable, "-m", "tests.dp_bucketed_worker"], env=env, cwd=ROOT, capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    for r in (8, 40):
        ranges = rec[f"r{r}_ranges"]
        assert len(ranges) >= 2 and ranges[0][0] == 0 and ranges[-1][1] == rec[f"r{r}_n_lora"]
        assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))      # the buckets tile the LoRA gradients
        # overlapped exchange: early buckets tile the up-path head of the buffer, late buckets the rest + the mapper gradient
        ov = rec[f"r{r}_overlap_ranges"]
        assert ov[0][0] == 0 and ov[-1][1] == rec[f"r{r}_numel"] and all(a[1] == b[0] for a, b in zip(ov, ov[1:]))
        assert any(a[1] == rec[f"r{r}_n_early"] for a in ov) and 0 < rec[f"r{r}_n_early"] < rec[f"r{r}_n_lora"]
        # three backward legs end with a hook-driven exchange: every cut of the bank is a bucket boundary, in increasing order
        cuts = rec[f"r{r}_cuts"]
        assert len(cuts) == 3 and 0 < cuts[0] < cuts[1] < cuts[2] < rec[f"r{r}_n_lora"], cuts
        assert all(any(a[1] == c for a in ov) for c in cuts), (cuts, ov)
        assert rec[f"r{r}_hook_buckets"] >= 3
        assert rec[f"r{r}_overlap_graphs"] == 1                            # collectives captured: the step is ONE graph
        lp = rec[f"r{r}_plain_losses"]
        if r == 8:
            # rob-finetune decoder (config 5): gradient buckets all-reduced from hooks DURING backward, through torch.distributed and
            # through aql_comm_* on a forked stream: 4 buckets, launched in backward order, same parameters as without exchange
            for mode in ("dist", "comm"):
                assert rec[f"robft_{mode}_buckets"] == 4 and rec[f"robft_{mode}_hook_launch_order"] == [0, 1, 2, 3], rec
                # two AdamW steps amplify the decoder step's own run-to-run noise (fp32 atomics in its reductions; the first updates
                # are lr * sign(g)): the bound is that spread, measured in the same process, not an absolute number
                assert rec[f"robft_{mode}_param_relerr"] <= 3 * rec["robft_plain_rerun_param_relerr"] + 1e-5, rec
                assert all(abs(a - b) < 1e-3 * abs(a) for a, b in zip(rec["robft_plain_losses"], rec[f"robft_{mode}_losses"])), rec
        for mode in ("overlap_eager", "overlap_graph", "bucketed_eager", "bucketed_graph"):
            assert rec[f"r{r}_{mode}_param_relerr"] < 2e-3, rec
            assert all(abs(a - b) < 2e-3 * abs(a) for a, b in zip(lp, rec[f"r{r}_{mode}_losses"])), rec
