"""GPU probe: norms, GEGLU, attention fwd/bwd, fused LoRA linear fwd/bwd vs torch fp32 references."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from aqualora_amd import ops, _lib as L
dev = "cuda"; torch.manual_seed(0)
ok = True
def rnd(*s, scale=1.0): return (torch.randn(*s, device=dev) * scale).to(torch.bfloat16)
def relerr(a, b):
    a = a.float(); b = b.float(); return ((a - b).abs().max() / (b.abs().max() + 1e-9)).item()
def report(name, err, tol):
    global ok; good = err < tol and err == err; ok &= good
    print(f"{'PASS' if good else 'FAIL'} {name}: relerr={err:.3e} (tol {tol})", flush=True)
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(n): fn()
    t1.record(); torch.cuda.synchronize(); return t0.elapsed_time(t1) / n

# GroupNorm + SiLU
for (B, C, H, W, silu, eps) in [(2, 320, 64, 64, 1, 1e-5), (2, 640, 16, 16, 0, 1e-6), (3, 2560, 8, 8, 1, 1e-5), (2, 32, 4, 4, 1, 1e-5), (2, 960, 32, 32, 1, 1e-5),
                                # maps above 32x32: the row-slab form (ragged slabs, non-square, VAE-like widths)
                                (3, 640, 64, 64, 1, 1e-5), (2, 960, 64, 64, 0, 1e-6), (1, 320, 72, 88, 1, 1e-5), (1, 128, 160, 96, 1, 1e-6),
                                # whole-row statistics pass (gn_rowstats_kernel): 8 samples (64 slabs), 32x32 backward at 8 samples, ragged last slab
                                (8, 320, 64, 64, 1, 1e-5), (8, 640, 32, 32, 1, 1e-5), (5, 960, 40, 52, 1, 1e-5)]:
    x = (rnd(B, C, H, W) * 2 + 0.5).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    g = rnd(C) * 0.5 + 1; b = rnd(C) * 0.1
    y = ops.groupnorm_silu(x, g, b, eps, silu)
    xr = x.detach().float().requires_grad_(True)
    yr = F.group_norm(xr, 32, g.float(), b.float(), eps); yr = F.silu(yr) if silu else yr
    report(f"groupnorm fwd C{C} {H}x{W} silu{silu}", relerr(y, yr), 1.5e-2)
    dy = rnd(B, C, H, W).contiguous(memory_format=torch.channels_last)
    y.backward(dy); yr.backward(dy.float())
    report(f"groupnorm bwd", relerr(x.grad, xr.grad), 2e-2)
# skip-connection concat (one launch forward, one launch backward)
for (B, Ca, Cb, H, W) in [(2, 320, 320, 16, 16), (1, 640, 320, 9, 7), (3, 8, 24, 5, 5)]:
    a = rnd(B, Ca, H, W).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    b = rnd(B, Cb, H, W).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = ops.cat_channels(a, b)
    report(f"cat_channels fwd {Ca}+{Cb} {H}x{W}", 0.0 if torch.equal(y, torch.cat([a, b], 1)) else 1.0, 1e-9)
    dy = rnd(B, Ca + Cb, H, W).contiguous(memory_format=torch.channels_last)
    y.backward(dy)
    report("cat_channels bwd", 0.0 if (torch.equal(a.grad, dy[:, :Ca]) and torch.equal(b.grad, dy[:, Ca:])) else 1.0, 1e-9)
# LayerNorm
for (M, C) in [(4096, 320), (1000, 640), (77, 1280), (5, 64)]:
    x = (rnd(M, C) * 2 + 0.3).requires_grad_(True); g = rnd(C) * 0.5 + 1; b = rnd(C) * 0.1
    y = ops.layernorm(x, g, b); xr = x.detach().float().requires_grad_(True); yr = F.layer_norm(xr, (C,), g.float(), b.float())
    report(f"layernorm fwd M{M} C{C}", relerr(y, yr), 1.5e-2)
    dy = rnd(M, C); y.backward(dy); yr.backward(dy.float())
    report("layernorm bwd", relerr(x.grad, xr.grad), 2e-2)
# GEGLU
x = rnd(1000, 2560).requires_grad_(True); y = ops.geglu(x)
xr = x.detach().float().requires_grad_(True); h, gt = xr.chunk(2, -1); yr = h * F.gelu(gt)
report("geglu fwd", relerr(y, yr), 1e-2); dy = rnd(1000, 1280); y.backward(dy); yr.backward(dy.float())
report("geglu bwd", relerr(x.grad, xr.grad), 1.5e-2)
# attention
def attn_ref(q, k, v, H):
    B, Nq, C = q.shape; d = C // H
    qh = q.view(B, Nq, H, d).transpose(1, 2); kh = k.view(B, -1, H, d).transpose(1, 2); vh = v.view(B, -1, H, d).transpose(1, 2)
    o = torch.softmax(qh @ kh.transpose(-1, -2) * d ** -0.5, -1) @ vh
    return o.transpose(1, 2).reshape(B, Nq, C)
for (B, H, Nq, Nk, d) in [(2, 8, 256, 256, 40), (1, 8, 1024, 1024, 80), (2, 8, 64, 64, 160), (2, 8, 256, 77, 40), (1, 8, 100, 77, 160), (1, 2, 200, 300, 64), (1, 8, 4096, 4096, 40),
                           # rob-finetune samples 512..768 px (rob_enhance_finetune.py:1004-1005): up to 96x96 = 9216 tokens, non-square maps
                           (1, 2, 9216, 9216, 40), (1, 2, 6336, 6336, 40), (1, 2, 6336, 77, 40), (1, 2, 1584, 1584, 80),
                           # text-state attention with the key side resident: several owner blocks per wavefront (4 at d = 40, 2 at d = 80), ragged Q ranges, few keys
                           (8, 8, 4096, 77, 40), (8, 8, 2048, 77, 80), (3, 8, 1000, 77, 40), (2, 8, 300, 13, 80), (2, 8, 64, 80, 160)]:
    q = rnd(B, Nq, H * d).requires_grad_(True); k = rnd(B, Nk, H * d).requires_grad_(True); v = rnd(B, Nk, H * d).requires_grad_(True)
    o = ops.attention(q, k, v, H)
    qr, kr, vr = [t.detach().float().requires_grad_(True) for t in (q, k, v)]
    orf = attn_ref(qr, kr, vr, H)
    report(f"attn fwd B{B} H{H} Nq{Nq} Nk{Nk} d{d}", relerr(o, orf), 1.5e-2)
    do = rnd(B, Nq, H * d); o.backward(do); orf.backward(do.float())
    report("attn dq", relerr(q.grad, qr.grad), 2e-2); report("attn dk", relerr(k.grad, kr.grad), 2e-2); report("attn dv", relerr(v.grad, vr.grad), 2e-2)
# spike test for online softmax rescale
q = rnd(1, 256, 320); k = rnd(1, 256, 320); v = rnd(1, 256, 320)
k[0, 200, :40] = q[0, 5, :40] * 20
report("attn fwd spike", relerr(ops.attention(q, k, v, 8), attn_ref(q.float(), k.float(), v.float(), 8)), 1.5e-2)
# the shift-in-the-MFMA forward (round 4): a late key far above the first tile's maximum.  x3 stays on the fast path (probabilities up
# to ~2^25 against the first tile's integer shift), x20 overflows fp32 and must take the classic pass; both through backward (lse),
# at d = 40 (register-staged tiles) and d = 80 (LDS-DMA tiles), and a row block whose scores all sit far below zero
for d, H, mult in ((40, 8, 3.0), (40, 8, 20.0), (80, 4, 3.0), (80, 4, 20.0)):
    q = rnd(2, 512, H * d).requires_grad_(True); k = rnd(2, 512, H * d); v = rnd(2, 512, H * d).requires_grad_(True)
    with torch.no_grad():
        k[0, 300, :d] = q[0, 5, :d] * mult
        k[1, 70, d:2 * d] = q[1, 400, d:2 * d] * mult
        q[1, 100:164] -= 0    # placeholder rows: plain
    k = k.requires_grad_(True)
    o = ops.attention(q, k, v, H)
    qr, kr, vr = [t.detach().float().requires_grad_(True) for t in (q, k, v)]
    orf = attn_ref(qr, kr, vr, H)
    # AQL_ATTN_FOLD=2 (opt-in: the shift inside the S-product) pays a second bf16 rounding of q * scale: ~0.0002 |s| log2 units,
    # visible only at |s| ~ 100 (profiles/r04_attention_fold.txt) -- its bounds under the x20 spike are wider
    loose = os.environ.get("AQL_ATTN_FOLD") == "2" and mult >= 20
    report(f"attn fwd late spike x{mult:g} d{d}", relerr(o, orf), 3e-2 if loose else 1.5e-2)
    do = rnd(2, 512, H * d); o.backward(do); orf.backward(do.float())
    gt = 8e-2 if loose else 2e-2
    report("attn dq", relerr(q.grad, qr.grad), gt); report("attn dk", relerr(k.grad, kr.grad), gt); report("attn dv", relerr(v.grad, vr.grad), gt)
q = rnd(1, 256, 320); k = rnd(1, 256, 320); v = rnd(1, 256, 320)
q[0, :, :40] = 6.0; k[0, :, :40] = -6.0 + 0.1 * torch.randn(256, 40, device=dev).to(k.dtype)     # head 0: every score ~ -1440 * scale
report("attn fwd all scores far below zero", relerr(ops.attention(q, k, v, 8), attn_ref(q.float(), k.float(), v.float(), 8)), 1.5e-2)
# fused LoRA linear fwd/bwd
class Site: pass
for (B, N, K, Nout, r) in [(2, 256, 320, 320, 32), (2, 77, 768, 640, 8), (2, 64, 1280, 10240, 32), (1, 1024, 320, 320, 320),
                           (2, 256, 640, 640, 16), (2, 64, 1280, 1280, 64), (2, 77, 768, 320, 320)]:
    M = B * N
    x = rnd(M, K).requires_grad_(True); W = rnd(Nout, K, scale=K ** -0.5); bias = rnd(Nout); A = torch.randn(r, K, device=dev) / r; Bu = torch.randn(Nout, r, device=dev) * 0.05
    S = (torch.randn(B, r, device=dev) * 0.3 + 1).requires_grad_(True); res = rnd(M, Nout)
    pk = ops.PackedLinear(W, bias)
    st = Site(); st.rank = r; st.a16 = A.to(torch.bfloat16); st.at16 = st.a16.t().contiguous(); st.b16 = Bu.to(torch.bfloat16); st.bt16 = st.b16.t().contiguous()
    st.ga = torch.zeros(r, K, device=dev); st.gb = torch.zeros(Nout, r, device=dev)
    S16 = S.to(torch.bfloat16)
    y = ops.lora_linear(x, pk, st, S, S16, N, res)
    xr = x.detach().float().requires_grad_(True); Ar = st.a16.float().requires_grad_(True); Br = st.b16.float().requires_grad_(True); Sr = S16.detach().float().requires_grad_(True)
    T = xr @ Ar.T; yr = xr @ W.float().T + bias.float() + (T * Sr.repeat_interleave(N, 0)) @ Br.T + res.float()
    report(f"lora_linear fwd M{M} K{K} N{Nout} r{r}", relerr(y, yr), 1.5e-2)
    dy = rnd(M, Nout); y.backward(dy); yr.backward(dy.float())
    report("lora_linear dx", relerr(x.grad, xr.grad), 2e-2); report("lora_linear dA", relerr(st.ga, Ar.grad), 2e-2)
    report("lora_linear dB", relerr(st.gb, Br.grad), 2e-2); report("lora_linear dS", relerr(S.grad, Sr.grad), 3e-2)
print("--- timing ---")
for (B, H, N, d) in [(4, 8, 4096, 40), (4, 8, 1024, 80), (4, 8, 256, 160)]:
    q = rnd(B, N, H * d).requires_grad_(True); k = rnd(B, N, H * d).requires_grad_(True); v = rnd(B, N, H * d).requires_grad_(True)
    ms = timeit(lambda: ops.attention(q, k, v, H)); fl = 4 * B * H * N * N * d
    o = ops.attention(q, k, v, H); do = rnd(B, N, H * d)
    msb = timeit(lambda: o.backward(do, retain_graph=True))
    qh = q.detach().view(B, N, H, d).transpose(1, 2); 
    mst = timeit(lambda: F.scaled_dot_product_attention(qh, qh, qh))
    print(f"attn B{B} N{N} d{d}: fwd {ms*1e3:.0f} us {fl/ms/1e9:.1f} TF/s | bwd {msb*1e3:.0f} us {2.5*fl/msb/1e9:.1f} TF/s | torch sdpa fwd {mst*1e3:.0f} us")
for (B, C, HW) in [(4, 320, 4096), (4, 1280, 256)]:
    x = rnd(B, C, 64 if HW == 4096 else 16, 64 if HW == 4096 else 16).contiguous(memory_format=torch.channels_last); g = rnd(C); b = rnd(C)
    ms = timeit(lambda: ops.groupnorm_silu(x, g, b, 1e-5, 1)); by = x.numel() * 2 * 3
    print(f"groupnorm B{B} C{C} HW{HW}: {ms*1e3:.1f} us  {by/ms/1e6:.0f} GB/s (3 passes)")
x = rnd(16384, 320); g = rnd(320); b = rnd(320)
ms = timeit(lambda: ops.layernorm(x, g, b)); print(f"layernorm 16384x320: {ms*1e3:.1f} us {x.numel()*4/ms/1e6:.0f} GB/s")
print("ALL PASS" if ok else "SOME FAILED")
