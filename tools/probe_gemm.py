"""GPU probe: validates the MFMA GEMM family against torch (fp32 math on bf16-rounded inputs) and times it."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from aqualora_amd import _lib as L

dev = "cuda"
torch.manual_seed(0)
ws = torch.empty(64 << 20, dtype=torch.float32, device=dev)

def rnd(*s, scale=1.0):
    return (torch.randn(*s, device=dev) * scale).to(torch.bfloat16)

def relerr(a, b):
    a = a.float(); b = b.float()
    return ((a - b).abs().max() / (b.abs().max() + 1e-9)).item()

def gemm(A, B, A2=None, B2=None, bias=None, rowbias=None, rps=1, res=None, use_ws=True):
    M, K = A.shape; N = B.shape[0]
    C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    L.call("aql_gemm_bf16", L.ptr(A), A.stride(0), L.ptr(B), B.stride(0), M, N, K,
           L.ptr(A2), 0 if A2 is None else A2.stride(0), L.ptr(B2), 0 if B2 is None else B2.stride(0),
           0 if A2 is None else A2.shape[1], L.ptr(bias), L.ptr(rowbias), rps, L.ptr(res), 0 if res is None else res.stride(0),
           L.ptr(C), N, L.ptr(ws) if use_ws else None, ws.numel() * 4 if use_ws else 0, L.stream_ptr())
    return C

def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(n): fn()
    t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1) / n

ok = True
def report(name, err, tol):
    global ok
    good = err < tol
    ok &= good
    print(f"{'PASS' if good else 'FAIL'} {name}: relerr={err:.3e} (tol {tol})", flush=True)

# 1. plain NT GEMM, all tile configs, tails
for (M, N, K) in [(256, 128, 64), (300, 320, 328), (1024, 32, 320), (308, 640, 768), (4096, 1280, 1280), (4, 1280, 320), (256, 1280, 11520)]:
    A = rnd(M, K); B = rnd(N, K, scale=K ** -0.5)
    bias = rnd(N); res = rnd(M, N)
    ref = (A.float() @ B.float().T + bias.float()).to(torch.bfloat16).float() + res.float()
    for use_ws in (False, True):
        C = gemm(A, B, bias=bias, res=res, use_ws=use_ws)
        report(f"gemm_nt M{M} N{N} K{K} ws={use_ws}", relerr(C, ref), 1.5e-2)

# 2. dual-K (LoRA) + rowbias
M, N, K, r = 2048, 320, 320, 32
A = rnd(M, K); B = rnd(N, K, scale=K ** -0.5); A2 = rnd(M, r); B2 = rnd(N, r, scale=0.1)
rb = rnd(M // 512, N)
ref = (A.float() @ B.float().T + A2.float() @ B2.float().T).to(torch.bfloat16).float() + rb.float().repeat_interleave(512, 0)
report("gemm dualK+rowbias", relerr(gemm(A, B, A2, B2, rowbias=rb, rps=512), ref), 1.5e-2)

# 3. lora_down
for (M, K, r, rps) in [(4096, 320, 32, 1024), (308, 768, 8, 77), (2048, 1280, 320, 256)]:
    X = rnd(M, K); Ad = rnd(r, K, scale=K ** -0.5); S = rnd(M // rps, r)
    T = torch.empty(M, r, dtype=torch.bfloat16, device=dev); Ts = torch.empty_like(T)
    L.call("aql_lora_down", L.ptr(X), K, M, K, L.ptr(Ad), r, L.ptr(S), rps, L.ptr(T), L.ptr(Ts), None, None, L.stream_ptr())
    Tr = (X.float() @ Ad.float().T)
    report(f"lora_down T M{M} K{K} r{r}", relerr(T, Tr), 1.5e-2)
    report(f"lora_down Ts", relerr(Ts, T.float() * S.float().repeat_interleave(rps, 0)), 1e-2)

# 4. conv3x3 fwd (stride 1 / 2 / upsample) and bwd data
def conv_case(Bn, H, W, Cin, Cout, stride, ups):
    x = rnd(Bn, Cin, H, W); w = rnd(Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5); b = rnd(Cout)
    xin = F.interpolate(x.float(), scale_factor=2.0, mode="nearest") if ups else x.float()
    ref = F.conv2d(xin, w.float(), b.float(), stride=stride, padding=1)
    xh = x.permute(0, 2, 3, 1).contiguous(); wk = w.permute(0, 2, 3, 1).contiguous().view(Cout, 9 * Cin)
    Ho, Wo = ref.shape[2], ref.shape[3]
    y = torch.empty(Bn, Ho, Wo, Cout, dtype=torch.bfloat16, device=dev)
    L.call("aql_conv3x3_fwd", L.ptr(xh), Bn, H, W, Cin, L.ptr(wk), L.ptr(b), Cout, stride, ups, None, 0, None, L.ptr(y),
           L.ptr(ws), ws.numel() * 4, L.stream_ptr())
    report(f"conv3x3 fwd B{Bn} {H}x{W} {Cin}->{Cout} s{stride} u{ups}", relerr(y.permute(0, 3, 1, 2), ref), 1.5e-2)
    if not ups:
        dy = rnd(Bn, Cout, Ho, Wo)
        dx_ref = torch.nn.grad.conv2d_input((Bn, Cin, H, W), w.float(), dy.float(), stride=stride, padding=1)
        wt = w.permute(1, 2, 3, 0).contiguous().view(Cin, 9 * Cout)
        dyh = dy.permute(0, 2, 3, 1).contiguous()
        dx = torch.empty(Bn, H, W, Cin, dtype=torch.bfloat16, device=dev)
        L.call("aql_conv3x3_bwd_data", L.ptr(dyh), Bn, H, W, Cin, L.ptr(wt), Cout, stride, L.ptr(dx), L.ptr(ws), ws.numel() * 4, L.stream_ptr())
        report(f"conv3x3 bwd_data s{stride}", relerr(dx.permute(0, 3, 1, 2), dx_ref), 1.5e-2)

conv_case(2, 16, 16, 64, 128, 1, 0)
conv_case(2, 16, 12, 32, 64, 2, 0)
conv_case(1, 8, 8, 64, 64, 1, 1)
# the VAE's maps on row tiles (round 4): 128- / 256-pixel rows, 512-pixel rows as two half-row tiles (real halo columns), 128-wide column
# tiles, forward and backward-data; batch > 1 so that tile rows cross image boundaries; a non-square map
# (the picker takes them from 256 row tiles on: >= 65536 output pixels)
conv_case(4, 128, 128, 64, 128, 1, 0)
conv_case(2, 128, 256, 128, 256, 1, 0)
conv_case(2, 64, 512, 128, 128, 1, 0)
conv_case(4, 128, 128, 128, 320, 1, 0)
conv_case(2, 8, 8, 1280, 1280, 1, 0)
conv_case(1, 64, 64, 8, 320, 1, 0)
# 64-pixel-wide stride-1 maps with Cin % 64 == 0: the row-tile kernel (aql_conv_row.cuh) whenever the 256x160 tile is chosen -- forced
# by AQL_TILE=14 (tests/test_gpu_kernels.py), by itself at 8 x 64 x 64 x 320 outputs (one chip-wide round)
conv_case(2, 64, 64, 128, 160, 1, 0)
conv_case(1, 12, 64, 64, 328, 1, 0)
conv_case(8, 64, 64, 64, 320, 1, 0)
conv_case(2, 64, 64, 320, 320, 1, 0)     # forward and backward-data (flipped taps) on 64-wide maps
conv_case(2, 32, 32, 320, 640, 1, 0)     # 32-wide maps: 4 image rows per 128-pixel tile
conv_case(1, 6, 64, 320, 160, 1, 0)      # H not a multiple of 4
conv_case(4, 16, 16, 1280, 1280, 1, 0)   # 16-wide maps, split K (fp32 slabs + finalize)
conv_case(4, 32, 32, 640, 640, 1, 0)     # 32-wide maps, split K

# 5. TN gemm (weight grads)
for (M, P, Q) in [(4096, 320, 32), (1000, 32, 768), (2048, 640, 320), (308, 1280, 8)]:
    U = rnd(M, P); V = rnd(M, Q)
    C = torch.zeros(P, Q, dtype=torch.float32, device=dev)
    L.call("aql_gemm_tn_f32", L.ptr(U), P, L.ptr(V), Q, M, P, Q, 1.0, L.ptr(C), Q, L.stream_ptr())
    report(f"gemm_tn M{M} P{P} Q{Q}", relerr(C, U.float().T @ V.float()), 2e-3)
# 5b. wide-rank weight gradients (transpose + NT kernels), accumulating onto a non-zero C
from aqualora_amd import ops as _ops
for (M, P, Q) in [(8192, 640, 320), (16384, 320, 768), (16384, 320, 320), (4000, 320, 1280), (616, 320, 768)]:
    U = rnd(M, P); V = rnd(M, Q)
    C0 = torch.randn(P, Q, device=dev)
    C = C0.clone()
    _ops.gemm_tn_acc(U, V, C, 0.5)
    report(f"gemm_tn wide M{M} P{P} Q{Q}", relerr(C, C0 + 0.5 * (U.float().T @ V.float())), 2e-3)

# 6. timing
print("--- timing (bf16, random data) ---")
for (M, N, K) in [(16384, 320, 320), (16384, 2560, 320), (16384, 320, 1280), (4096, 640, 640), (4096, 5120, 640), (1024, 1280, 1280), (8192, 8192, 8192)]:
    A = rnd(M, K); B = rnd(N, K)
    ms = timeit(lambda: gemm(A, B))
    ms_t = timeit(lambda: A @ B.T)
    print(f"gemm M{M} N{N} K{K}: {ms*1e3:.1f} us  {2*M*N*K/ms/1e9:.1f} TF/s | torch {ms_t*1e3:.1f} us {2*M*N*K/ms_t/1e9:.1f} TF/s", flush=True)
for (Bn, H, Cin, Cout) in [(4, 64, 320, 320), (4, 32, 640, 640), (4, 16, 1280, 1280), (4, 8, 1280, 1280), (4, 16, 2560, 1280), (4, 64, 960, 320)]:
    xh = rnd(Bn, H, H, Cin); wk = rnd(Cout, 9 * Cin); b = rnd(Cout)
    y = torch.empty(Bn, H, H, Cout, dtype=torch.bfloat16, device=dev)
    fn = lambda: L.call("aql_conv3x3_fwd", L.ptr(xh), Bn, H, H, Cin, L.ptr(wk), L.ptr(b), Cout, 1, 0, None, 0, None, L.ptr(y), L.ptr(ws), ws.numel() * 4, L.stream_ptr())
    ms = timeit(fn)
    xn = xh.permute(0, 3, 1, 2); wn = wk.view(Cout, 3, 3, Cin).permute(0, 3, 1, 2).contiguous(memory_format=torch.channels_last)
    ms_t = timeit(lambda: F.conv2d(xn, wn, b, padding=1))
    fl = 2 * Bn * H * H * Cout * 9 * Cin
    print(f"conv3x3 B{Bn} {H}x{H} {Cin}->{Cout}: {ms*1e3:.1f} us {fl/ms/1e9:.1f} TF/s | miopen {ms_t*1e3:.1f} us {fl/ms_t/1e9:.1f} TF/s", flush=True)
print("ALL PASS" if ok else "SOME FAILED")
