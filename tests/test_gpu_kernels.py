"""GPU: per-kernel parity sweeps (GEMM family, conv fwd/bwd, norms, GEGLU, attention fwd/bwd incl. ragged / cross /
spiked-softmax cases, fused LoRA linear at ranks 8..320) against fp32 torch references of the same op.  The sweeps live
in tools/probe_gemm.py and tools/probe_ops.py (they also print timings); each prints PASS/FAIL per case with its
tolerance and a final verdict."""
import os
import subprocess
import sys

import pytest
import torch

from tests.conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(script, env=None):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", script)], capture_output=True, text=True,
                         timeout=900, env=None if env is None else {**os.environ, **env})
    text = out.stdout + out.stderr
    fails = [l for l in text.splitlines() if l.startswith("FAIL")]
    assert out.returncode == 0 and "ALL PASS" in text and not fails, "\n".join(fails[:20]) or text[-2000:]
    return text


def test_gemm_family_parity():
    text = _run("probe_gemm.py")
    assert text.count("PASS") >= 30


def test_gemm_family_parity_on_the_12_wave_256x160_kernel():
    """The picker takes the 256x160 kernel only for grids of one chip-wide round (the batch-4 twin forward); AQL_TILE=14 forces
    it on every 160-wide shape of the sweep (ragged M, split-K, conv forward / backward-data, residual / row-bias epilogues)."""
    text = _run("probe_gemm.py", {"AQL_TILE": "14"})
    assert text.count("PASS") >= 30


def test_gemm_family_parity_on_the_8_wave_128x160_kernel():
    """AQL_TILE=11 forces the wave-specialised 128x160 tile: the stride-1 convolutions of the sweep then run the 128-row row-tile
    kernels (64- / 32- / 16-pixel-wide maps, forward and flipped-tap backward-data, with and without split K)."""
    text = _run("probe_gemm.py", {"AQL_TILE": "11"})
    assert text.count("PASS") >= 30


def test_lora_geglu_256x256_persistent_tile_is_bit_identical():
    """aql_gemm_lora_t256.cuh (the default of aql_lora_gemm_fused_geglu from 512 tiles on) forced on every GEGLU form it can run
    (AQL_LORA_CFG=t256) against the 128 x 160 one-shot kernel: G / H / T / Ts equal bit for bit, ragged rows, K tails, twin row0."""
    text = _run("probe_lora_persist.py", {"PROBE_ALT": "t256"})
    assert text.count("PASS geglu") >= 13


def test_geglu_gemm_with_a_second_k_segment_on_the_256x256_tile_is_bit_identical():
    """aql_gemm_bf16_geglu (rank != 32: the LoRA term as a second K segment on rows >= row0; or no LoRA) on the SEG2 form of the
    256 x 256 tile against the 128 x 160 kernels: G / H equal bit for bit, NaN-poisoned Ts rows below row0 never read."""
    text = _run("probe_t256_seg2.py")
    assert text.count("PASS") >= 12


def test_persistent_256x256_kernel_repeats_bit_for_bit():
    """tools/stress_t256.py: the same launch 150 times on NaN-poisoned outputs, every result equal to the first (a missed barrier or
    vmcnt wait of the hand-phased pipeline would show up as a difference)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress_t256.py"), "150"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "ALL PASS" in out.stdout, (out.stdout + out.stderr)[-1500:]


def test_geglu_epilogue_on_the_256x256_tile_vs_fp32():
    text = _run("probe_geglu.py", {"AQL_LORA_CFG": "t256"})
    assert text.count("PASS") >= 20


def test_ops_parity():
    text = _run("probe_ops.py")
    assert text.count("PASS") >= 93   # incl. the shift-in-the-MFMA forward under late spikes (fast path and overflow fallback), attention at 9216 / 6336 tokens (768 px and non-square rob-finetune samples)


@pytest.mark.parametrize("env", [{"AQL_ATTN_FOLD": "0", "AQL_ATTN_DFOLD": "0"}])
def test_ops_parity_on_the_other_attention_loops(env):
    """The running-maximum forward (also the overflow fallback of the default loop) with the per-element `dP - delta` backward: the
    same sweep as test_ops_parity.  (The shelved forward with the shift inside the S-product, AQL_ATTN_FOLD=2, left the product build
    in round 5: it exists in -DAQL_EXPERIMENTS builds only, tools/build_alt.sh.)"""
    text = _run("probe_ops.py", env)
    assert text.count("PASS") >= 93


def test_self_attention_on_a_prescaled_q_vs_fp32():
    """aql_sdpa_fwd_qpre / aql_sdpa_bwd_qpre (q multiplied by d^-1/2 log2(e) by its producer; the softmax shift inside the S-product in
    all three passes) against an fp32 softmax attention on the operands the kernels see: 256- and 128-row workgroups, a late key 3x and
    20x above the first tile's scores (fast path / overflow fallback), ragged N -- o, dq (gradient of the UNSCALED q), dk, dv
    (tools/probe_attn_qpre.py); the LFOLD-off backward (AQL_ATTN_LFOLD=0) through the same sweep."""
    assert _run("probe_attn_qpre.py").count("PASS attention qpre") == 6
    assert _run("probe_attn_qpre.py", {"AQL_ATTN_LFOLD": "0"}).count("PASS attention qpre") == 6


def test_transpose_read_weight_gradient_gemm():
    """aql_gemm_tn_tr_f32 (wide 128x128 and rank <= 32 128x32 tiles, swapped / transposed output, ragged M, P, Q, strided
    operands) against fp32 torch."""
    text = _run("probe_tntr.py")
    assert text.count("PASS") >= 25


def test_transpose_read_weight_gradient_gemm_dma_ring():
    """The LDS-DMA ring body of the same kernel (AQL_TNTR_NST=2: kept as an option, faster alone, slower inside the step)."""
    text = _run("probe_tntr.py", {"AQL_TNTR_NST": "2"})
    assert text.count("PASS") >= 25


def test_one_launch_lora_linear():
    """aql_lora_gemm_fused (T side accumulator + up-projection k-step, 4-wave and wave-specialised kernels) vs fp32 torch."""
    text = _run("probe_lora_gemm.py")
    assert text.count("PASS") >= 17   # incl. the grouped launch (q|k|v, text-state k|v of all blocks) == separate launches


def test_geglu_epilogue_equals_two_kernel_path():
    """aql_gemm_bf16_geglu / aql_lora_gemm_fused_geglu: G and H bit-identical to GEMM + aql_geglu_fwd on the U-Net's ff.net.0
    shapes (rank 0 / 32 / 8, ragged M, tiny F), within 1.5e-2 of fp32 torch, and the same gradients through autograd."""
    text = _run("probe_geglu.py")
    assert text.count("PASS") >= 26   # incl. the GEGLU backward in the ff.net.2 backward-data epilogue (ops.FeedForwardFn)


def test_lora_down_splitk_equals_one_piece_sum():
    """aql_lora_down_splitk (deep K under few rows: K range cut over workgroups, pieces added in piece order by the finalize launch) against fp32
    torch and against aql_lora_down on the backward-data shapes of ff.net.0 at the 16x16 / 8x8 levels, a ragged row count and a K
    that is not a multiple of the piece size; repeated launches reuse the self-resetting counters and give identical bits."""
    from aqualora_amd import _lib as L
    torch.manual_seed(1)
    cnt = torch.zeros(4096, dtype=torch.int32, device="cuda")
    ws = torch.empty(32 * 2048 * 32, dtype=torch.float32, device="cuda")
    for M, K in ((1024, 10240), (256, 10240), (1024, 5120), (154, 2080), (2048, 5120), (16, 4096)):
        X = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        A = (torch.randn(32, K, device="cuda") / 32).to(torch.bfloat16)
        S = torch.randn(2, 32, device="cuda").to(torch.bfloat16)
        outs = []
        for rep in range(3):
            T = torch.full((M, 32), float("nan"), device="cuda", dtype=torch.bfloat16)
            Ts = T.clone()
            ws.fill_(float("nan"))
            L.call("aql_lora_down_splitk", L.ptr(X), K, M, K, L.ptr(A), 32, L.ptr(S), M // 2, L.ptr(T), L.ptr(Ts), L.ptr(ws),
                   ws.numel() * 4, L.ptr(cnt), cnt.numel() * 4, L.stream_ptr())
            outs.append((T, Ts))
        assert int(cnt.abs().max()) == 0, (M, K)
        T, Ts = outs[0]
        ref = X.float() @ A.float().t()
        assert torch.isfinite(T.float()).all() and torch.isfinite(Ts.float()).all(), (M, K)
        assert float((T.float() - ref).abs().max() / ref.abs().max()) < 1e-2, (M, K)
        ts_ref = T.float() * S.float().repeat_interleave(M // 2, dim=0)
        assert float((Ts.float() - ts_ref).abs().max() / ts_ref.abs().max()) < 1e-2, (M, K)
        for T2, Ts2 in outs[1:]:
            assert torch.equal(T2, T) and torch.equal(Ts2, Ts), (M, K)
        T1 = torch.empty_like(T)
        Ts1 = torch.empty_like(T)
        L.call("aql_lora_down", L.ptr(X), K, M, K, L.ptr(A), 32, L.ptr(S), M // 2, L.ptr(T1), L.ptr(Ts1), None, None, L.stream_ptr())
        # different summation order in fp32, then one bf16 rounding: equal up to a bf16 ulp at rounding boundaries
        assert float((T.float() - T1.float()).abs().max() / ref.abs().max()) < 8e-3, (M, K)


def test_lora_down_splitk_ticket_form_stress():
    """The opt-in one-launch form of aql_lora_down_splitk (AQL_DOWN_TICKET=1: partials exchanged between workgroups of different XCDs
    through sc1 stores / loads and a relaxed ticket, no fence -- see the kernel's comment) against the default two-launch form
    (ordered by the kernel boundary): 2000 launches on four rotating inputs over scratch that still holds the PREVIOUS input's
    partials, with a bandwidth-hungry copy running beside them; a stale or torn piece would change bits.  Every output must equal
    the two-launch form's bit for bit, and the counters must be left zero."""
    import os
    from aqualora_amd import _lib as L
    torch.manual_seed(7)
    cnt = torch.zeros(4096, dtype=torch.int32, device="cuda")
    ws = torch.full((32 * 2048 * 32,), float("nan"), dtype=torch.float32, device="cuda")
    big = torch.randn(64 << 20, device="cuda")
    big2 = torch.empty_like(big)
    side = torch.cuda.Stream()
    old = os.environ.pop("AQL_DOWN_TICKET", None)
    try:
        for M, K in ((1024, 10240), (256, 10240)):
            Xs = [torch.randn(M, K, device="cuda").to(torch.bfloat16) for _ in range(4)]
            A = (torch.randn(32, K, device="cuda") / 32).to(torch.bfloat16)
            S = torch.randn(2, 32, device="cuda").to(torch.bfloat16)

            def run(X):
                T = torch.full((M, 32), float("nan"), device="cuda", dtype=torch.bfloat16)
                Ts = T.clone()
                L.call("aql_lora_down_splitk", L.ptr(X), K, M, K, L.ptr(A), 32, L.ptr(S), M // 2, L.ptr(T), L.ptr(Ts), L.ptr(ws),
                       ws.numel() * 4, L.ptr(cnt), cnt.numel() * 4, L.stream_ptr())
                return T, Ts

            os.environ.pop("AQL_DOWN_TICKET", None)
            want = [run(X) for X in Xs]
            assert all(torch.isfinite(t.float()).all() for pair in want for t in pair)
            assert not torch.equal(want[0][0], want[1][0])
            os.environ["AQL_DOWN_TICKET"] = "1"
            bad = torch.zeros((), dtype=torch.int64, device="cuda")
            for i in range(2000):
                if i % 50 == 0:
                    with torch.cuda.stream(side):
                        big2.copy_(big)
                T, Ts = run(Xs[i % 4])
                bad += (T.view(torch.int16) != want[i % 4][0].view(torch.int16)).sum() + \
                    (Ts.view(torch.int16) != want[i % 4][1].view(torch.int16)).sum()
            torch.cuda.synchronize()
            assert int(bad) == 0, (M, K, int(bad))
            assert int(cnt.abs().max()) == 0
    finally:
        os.environ.pop("AQL_DOWN_TICKET", None)
        if old is not None:
            os.environ["AQL_DOWN_TICKET"] = old


def test_lora_down_skinny_every_k_step_count():
    """aql_lora_down (rank 32: the skinny MFMA kernel) for every K step count 1..12 and a ragged row count, with and without the
    fused dS reduction, against fp32 torch.  K < 256 leaves some wavefronts without a second K step: until round 3 their MFMAs ran
    anyway under an EXEC mask (MFMA ignores EXEC) and returned NaN -- never reached by the U-Net's K >= 320."""
    from aqualora_amd import _lib as L
    torch.manual_seed(0)
    for M in (154, 512):
        for K in range(32, 32 * 13, 32):
            X = torch.randn(M, K, device="cuda").to(torch.bfloat16)
            A = (torch.randn(32, K, device="cuda") / 8).to(torch.bfloat16)
            S = torch.randn(2, 32, device="cuda").to(torch.bfloat16)
            Tref = torch.randn(M, 32, device="cuda").to(torch.bfloat16)
            T = torch.full((M, 32), float("nan"), device="cuda", dtype=torch.bfloat16)
            Ts, dS = T.clone(), torch.zeros(2, 32, device="cuda")
            L.call("aql_lora_down", L.ptr(X), K, M, K, L.ptr(A), 32, L.ptr(S), M // 2, L.ptr(T), L.ptr(Ts), L.ptr(Tref), L.ptr(dS),
                   L.stream_ptr())
            ref = X.float() @ A.float().t()
            assert torch.isfinite(T.float()).all() and torch.isfinite(Ts.float()).all(), (M, K)
            assert float((T.float() - ref).abs().max() / ref.abs().max()) < 1e-2, (M, K)
            ts_ref = T.float() * S.float().repeat_interleave(M // 2, dim=0)
            assert float((Ts.float() - ts_ref).abs().max() / ts_ref.abs().max()) < 1e-2, (M, K)
            ds_ref = (T.float() * Tref.float()).view(2, M // 2, 32).sum(1)
            assert float((dS - ds_ref).abs().max() / ds_ref.abs().max()) < 1e-4, (M, K)


def test_splitk_finalize_inside_the_groupnorm_is_bit_identical():
    """Round 6 (tools/probe_defer.py): split-K 3x3 convolutions whose finalize launch is left to the GroupNorm behind them
    (aql_conv3x3_fwd_defer / _bwd_data_defer + aql_groupnorm_silu_fwd_slabs / _bwd_slabs) against the launch pairs they replace --
    conv output, normalised map, statistics, GroupNorm input gradient BIT for bit on the U-Net's 8x8 .. 32x32 maps at batch 1 - 8,
    with bias / row bias / residual epilogues; ResnetBlock2D forward + backward with the deferral on == off (twin and plain);
    an unconsumed deferred output is finished by the next user of the slab buffer."""
    text = _run("probe_defer.py")
    assert text.count("PASS") >= 40


def test_row_resident_chain_kernel_is_bit_identical_to_the_launch_sequence():
    """aql_lora_chain_fwd (csrc/aql_chain.hip) against aql_lora_gemm_fused (+ residual) -> aql_layernorm_fwd -> aql_lora_gemm_fused x n on
    the 64 x 64 level's shapes (twin batch, plain batch, two chip-wide rounds): hs, LayerNorm output, statistics, q / k / v, T / Ts of
    every linear equal bit for bit (tools/probe_chain.py)."""
    text = _run("probe_chain.py")
    assert text.count("PASS chain") >= 5


def test_row_resident_chain_kernel_repeats_bit_for_bit_under_memory_load():
    """tools/stress_chain2.py: 8 chain launches back to back behind a bandwidth-bound kernel, 20 times per chain form, every output equal
    to the first run.  Round 5 found this way that a 16-byte buffer store with an SGPR soffset reads its data registers late when the
    memory pipeline is backed up (hipcc inserts no hazard wait): one row in ~1 launch of 5, never on an idle chip."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress_chain2.py"), "20"], capture_output=True, text=True, timeout=900)
    text = out.stdout + out.stderr
    lines = [l for l in text.splitlines() if "mismatching outputs" in l]
    assert out.returncode == 0 and len(lines) == 3 and all("mismatching outputs: 0 " in l for l in lines), text[-2000:]


def test_rank320_chain_kernel_repeats_bit_for_bit_under_memory_load():
    """The same stress on the rank-320 kernel (aql_lora_chain_fwd_r320, config 3's 65536-row twin batch): 4 launches back to back behind
    a bandwidth-bound kernel, 10 times per chain form."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress_chain2.py"), "10", "wide"], capture_output=True, text=True, timeout=900)
    text = out.stdout + out.stderr
    lines = [l for l in text.splitlines() if "mismatching outputs" in l]
    assert out.returncode == 0 and len(lines) == 3 and all("mismatching outputs: 0 " in l for l in lines), text[-2000:]


@pytest.mark.parametrize("rank", [32, 320])
def test_transformer_block_through_chains_equals_the_per_launch_block(rank):
    """unet.Transformer2DModel at 320 channels on a twin batch of 8 x 64 x 64: with ops.CHAIN the three chains (ops.ChainFn) replace nine
    launches (rank 320, BASELINE config 3: sixteen -- aql_lora_chain_fwd_r320 also holds the down products).  Forward output bit-identical; the input gradient and every LoRA weight gradient as close as two runs of the per-launch
    path are to each other (the weight-gradient GEMMs accumulate with fp32 atomics)."""
    from aqualora_amd import ops, synth
    from aqualora_amd.lora import inject_lora
    from aqualora_amd.unet import Transformer2DModel
    torch.manual_seed(0)
    dev = "cuda"
    B, C, H = 4, 320, 64
    tm = Transformer2DModel(C, 8, 768, device=dev, dtype=torch.bfloat16)
    with torch.no_grad():
        for n_, p_ in tm.named_parameters():
            if p_.dim() >= 2:
                p_.copy_(synth.normal(n_, p_.shape, p_[0].numel() ** -0.5, 7, dev).to(p_.dtype))
            elif n_.endswith("bias"):
                p_.copy_(synth.normal(n_, p_.shape, 0.05, 7, dev).to(p_.dtype))
    for p_ in tm.parameters():
        p_.requires_grad_(False)
    keys = [n_ for n_, m in tm.named_modules() if hasattr(m, "lora_layer") and (n_.startswith("proj") or "attn" in n_ or "ff" in n_)]
    inject_lora(tm, rank, keys)
    with torch.no_grad():
        for k in keys:
            lay = tm.get_submodule(k).lora_layer
            lay.down.weight.copy_(synth.normal(k + ".d", lay.down.weight.shape, 1.0 / rank, 7, dev))
            lay.up.weight.copy_(synth.normal(k + ".u", lay.up.weight.shape, 0.05, 7, dev))
    lparams = [p_ for k in keys for p_ in (tm.get_submodule(k).lora_layer.down.weight, tm.get_submodule(k).lora_layer.up.weight)]
    x0 = synth.normal("x", (2 * B, C, H, H), 1.0, 7, dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    ctx0 = synth.normal("ctx", (2 * B, 77, 768), 1.0, 7, dev).to(torch.bfloat16)
    S0 = torch.cat([torch.zeros(B, rank, device=dev), synth.normal("S", (B, rank), 1.0, 7, dev)])
    dy = synth.normal("dy", (B, C, H, H), 1.0, 7, dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)

    def run(chain, qpre=False):
        ops.CHAIN, ops.QPRE = chain, qpre
        for p_ in lparams:
            p_.grad = None
        ops.dual_begin()
        try:
            x = ops.make_twin(x0[:B], x0[B:]).requires_grad_(True)
            ctx = ops.make_twin(ctx0[:B], ctx0[B:])
            S = S0[B:].clone().requires_grad_(True)
            S16f = S0.to(torch.bfloat16).contiguous()
            ops.DUAL.register(S16f)
            S._aql_s16 = S16f[B:]
            # k | v of the text states as UNet.forward provides them to the cross-attention
            a2 = tm.transformer_blocks[0].attn2
            k2 = a2.to_k(ctx, S)
            v2 = a2.to_v(ctx, S)
            ctx._aql_kv = {id(a2): (k2, v2)}
            y = tm(x, ctx, S)
            yk = ops._full(y).detach().clone()
            y.backward(dy)
            torch.cuda.synchronize()
            return yk, x.grad.detach().clone(), torch.cat([p_.grad.reshape(-1).float() for p_ in lparams]), S.grad.detach().clone()
        finally:
            ops.dual_end()
            ops.CHAIN, ops.QPRE = True, True

    y_a, dx_a, g_a, ds_a = run(False)
    y_b, dx_b, g_b, ds_b = run(False)
    y_c, dx_c, g_c, ds_c = run(True)
    assert torch.equal(y_a, y_b) and torch.equal(y_a, y_c)              # forward: bit-identical (both halves of the twin batch)
    # backward-data: the same launches except where this bank-less set-up sums the three q | k | v input gradients differently
    # (autograd adds three bf16 tensors; the chain's backward adds them in the GEMM epilogues, as the trainer's grouped path does)
    assert torch.equal(dx_a, dx_b)
    e_dx = ((dx_a.float() - dx_c.float()).norm() / dx_a.float().norm()).item()
    assert e_dx < 3e-3, e_dx
    spread = ((g_a - g_b).abs().max() / g_a.abs().max()).item()
    diff = ((g_a - g_c).abs().max() / g_a.abs().max()).item()
    print(f"chains vs per-launch block: dx l2rel {e_dx:.2e}, weight gradients max-rel {diff:.2e} (two per-launch runs: {spread:.2e})")
    # upstream of the q | k | v sum the gradients inherit its bf16-level difference (proj_in, norm1); everything else sees equal inputs
    assert g_a.abs().max() > 0 and diff < 5e-3, (spread, diff)
    assert ((ds_a - ds_c).abs().max() / ds_a.abs().max()).item() < 5e-3
    # the default form of the chains: attn1.to_q leaves the chain multiplied by d^-1/2 log2(e) and the self-attention runs on
    # aql_sdpa_fwd_qpre / _bwd_qpre (the softmax shift inside the S-product) -- q is ROUNDED at another point, so not the same bits
    y_d, dx_d, g_d, ds_d = run(True, qpre=True)
    e_y = ((y_a.float() - y_d.float()).norm() / y_a.float().norm()).item()
    e_dx = ((dx_a.float() - dx_d.float()).norm() / dx_a.float().norm()).item()
    diff = ((g_a - g_d).abs().max() / g_a.abs().max()).item()
    print(f"pre-scaled q: y l2rel {e_y:.2e}, dx l2rel {e_dx:.2e}, weight gradients max-rel {diff:.2e}")
    assert e_y < 8e-3 and e_dx < 1e-2 and diff < 1.5e-2 and ((ds_a - ds_d).abs().max() / ds_a.abs().max()).item() < 1e-2


@pytest.mark.parametrize("C,H", [(640, 32), (1280, 16), (320, 32)])
def test_rank320_grouped_qkv_equals_the_per_site_launches(C, H):
    """BASELINE config 3 (rank 320) at the 640- / 1280-channel levels (and a 320-channel map below the chains' tile count): q | k | v of
    the self-attention through ops.GroupedWideFn -- one stacked down product, one column-grouped GEMM (aql_gemm_bf16_grouped), the
    attention backward writing [dQ | dK | dV] as one buffer (aql_sdpa_bwd_ex), one grouped backward down product, one K-concatenated
    dX GEMM -- against the twelve per-site launches (AQL_GROUPED_WIDE=0), on a twin batch through the trainer's deferred weight-gradient
    machinery: block output BIT-identical; input gradient, the LoRA weight gradients and dS equal up to dX's single fp32 accumulation
    (the per-site path rounds the partial sums of q, k and v to bf16 on the way)."""
    from aqualora_amd import ops, synth
    from aqualora_amd.lora import LoraBank, inject_lora
    from aqualora_amd.unet import Transformer2DModel
    torch.manual_seed(0)
    dev, rank, B = "cuda", 320, 4
    tm = Transformer2DModel(C, 8, 768, device=dev, dtype=torch.bfloat16)
    with torch.no_grad():
        for n_, p_ in tm.named_parameters():
            if p_.dim() >= 2:
                p_.copy_(synth.normal(n_, p_.shape, p_[0].numel() ** -0.5, 7, dev).to(p_.dtype))
            elif n_.endswith("bias"):
                p_.copy_(synth.normal(n_, p_.shape, 0.05, 7, dev).to(p_.dtype))
    for p_ in tm.parameters():
        p_.requires_grad_(False)
    keys = [n_ for n_, m in tm.named_modules() if hasattr(m, "lora_layer") and (n_.startswith("proj") or "attn" in n_ or "ff" in n_)]
    inject_lora(tm, rank, keys)
    with torch.no_grad():
        for k in keys:
            lay = tm.get_submodule(k).lora_layer
            lay.down.weight.copy_(synth.normal(k + ".d", lay.down.weight.shape, 1.0 / rank, 7, dev))
            lay.up.weight.copy_(synth.normal(k + ".u", lay.up.weight.shape, 0.05, 7, dev))
    bank = LoraBank(tm, keys=keys)
    x0 = synth.normal("x", (2 * B, C, H, H), 1.0, 7, dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    ctx0 = synth.normal("ctx", (2 * B, 77, 768), 1.0, 7, dev).to(torch.bfloat16)
    S0 = torch.cat([torch.zeros(B, rank, device=dev), synth.normal("S", (B, rank), 1.0, 7, dev)])
    dy = synth.normal("dy", (B, C, H, H), 1.0, 7, dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)

    def run(wide):
        ops.GROUPED_WIDE = wide
        bank.zero_grad()
        ds_accum = torch.zeros(B, rank, dtype=torch.float32, device=dev)
        ops.DEFERRED = ops.DeferredDW(torch.device(dev, 0))
        ops.dual_begin()
        try:
            x = ops.make_twin(x0[:B], x0[B:]).requires_grad_(True)
            ctx = ops.make_twin(ctx0[:B], ctx0[B:])
            S = S0[B:].clone().requires_grad_(True)
            S16f = S0.to(torch.bfloat16).contiguous()
            ops.DUAL.register(S16f)
            S._aql_s16 = S16f[B:]
            S._aql_ds_accum = ds_accum
            y = tm(x, ctx, S)
            yk = ops._full(y).detach().clone()
            y.backward(dy)
            ops.DEFERRED.flush()
            ops.fold_ds3(ds_accum)
            torch.cuda.synchronize()
            return yk, x.grad.detach().clone(), bank.grad[:bank.numel].clone(), ds_accum.clone()
        finally:
            ops.dual_end()
            ops.DEFERRED = None
            ops.GROUPED_WIDE = True

    y_a, dx_a, g_a, ds_a = run(False)
    y_b, dx_b, g_b, ds_b = run(False)
    y_c, dx_c, g_c, ds_c = run(True)
    assert torch.isfinite(y_c.float()).all() and torch.isfinite(g_c).all()
    assert torch.equal(y_a, y_b) and torch.equal(y_a, y_c)              # forward: bit-identical (both halves of the twin batch)
    e_dx = ((dx_a.float() - dx_c.float()).norm() / dx_a.float().norm()).item()
    spread = ((g_a - g_b).abs().max() / g_a.abs().max()).item()
    diff = ((g_a - g_c).abs().max() / g_a.abs().max()).item()
    l2 = ((g_a - g_c).norm() / g_a.norm()).item()
    e_ds = ((ds_a - ds_c).abs().max() / ds_a.abs().max()).item()
    print(f"C={C}: grouped rank-320 q|k|v vs per-site: dx l2rel {e_dx:.2e}, weight gradients max-rel {diff:.2e} l2rel {l2:.2e} "
          f"(two per-site runs: {spread:.2e}), dS max-rel {e_ds:.2e}")
    assert e_dx < 4e-3, e_dx
    assert g_a.abs().max() > 0 and diff < 1e-2 and l2 < 4e-3, (spread, diff, l2)
    assert e_ds < 5e-3, e_ds


@pytest.mark.parametrize("B,N", [(4, 4096), (1, 2048)])   # 256-row (NOF = 4) and 128-row (NOF = 2) workgroups of attn_fwd_kernel
def test_attention_overflow_fallback_forward_and_backward_vs_fp32(B, N):
    """The default self-attention forward fixes the softmax shift from the FIRST key tile and re-runs a workgroup with the
    running-maximum pass when a later score overflows it (csrc/aql_attn.hip, FOLD = 1).  Here the LAST key scores ~150 (natural log
    units) above everything in the first tile for every second query row, so the fallback runs in every workgroup while half of each
    workgroup's rows would have been fine: O, and dq / dk / dv through the saved lse, against an fp32 softmax attention."""
    from aqualora_amd import ops
    torch.manual_seed(5)
    dev, H, d = "cuda", 8, 40
    C = H * d
    q = torch.randn(B, N, H, d, device=dev)
    k = torch.randn(B, N, H, d, device=dev)
    v = torch.randn(B, N, H, d, device=dev)
    u = torch.ones(d, device=dev) / d ** 0.5
    q[:, ::2] += 4.0 * d ** 0.5 * u          # q.u = 4 sqrt(d) + noise on the even rows
    k[:, -1] = 40.0 * u                      # => score (q.k) / sqrt(d) ~ 160 on those rows, ~0 on the odd ones
    q16, k16, v16 = (t.reshape(B, N, C).to(torch.bfloat16).requires_grad_(True) for t in (q, k, v))
    do = torch.randn(B, N, C, device=dev).to(torch.bfloat16)
    o = ops.attention(q16, k16, v16, H)
    o.backward(do)
    assert torch.isfinite(o).all() and all(torch.isfinite(t.grad).all() for t in (q16, k16, v16))
    # fp32 reference on the bf16-rounded inputs, a few heads at a time
    qf, kf, vf = (t.detach().float().view(B, N, H, d).permute(0, 2, 1, 3).requires_grad_(True) for t in (q16, k16, v16))
    dof = do.float().view(B, N, H, d).permute(0, 2, 1, 3)
    outs = []
    for b in range(B):
        s = torch.einsum("hqd,hkd->hqk", qf[b], kf[b]) * d ** -0.5
        ob = torch.einsum("hqk,hkd->hqd", torch.softmax(s, dim=-1), vf[b])
        ob.backward(dof[b])
        outs.append(ob.detach())
        del s, ob
    of = torch.stack(outs).permute(0, 2, 1, 3).reshape(B, N, C)
    err = lambda a, b_: ((a.float() - b_).abs().max() / b_.abs().max()).item()   # noqa: E731
    e_o = err(o, of)
    e_q = err(q16.grad, qf.grad.permute(0, 2, 1, 3).reshape(B, N, C))
    e_k = err(k16.grad, kf.grad.permute(0, 2, 1, 3).reshape(B, N, C))
    e_v = err(v16.grad, vf.grad.permute(0, 2, 1, 3).reshape(B, N, C))
    print(f"attention under a late spike, B {B} N {N}: o {e_o:.2e} dq {e_q:.2e} dk {e_k:.2e} dv {e_v:.2e}")
    assert e_o < 8e-3 and e_q < 1.5e-2 and e_k < 1.5e-2 and e_v < 1.5e-2, (e_o, e_q, e_k, e_v)
