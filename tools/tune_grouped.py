"""Round 6: the rank-320 grouped launches (aql_gemm_bf16_grouped / aql_lora_down on the stacked A / the K-concatenated dX GEMM) on the
shapes config 3 runs them, HIP-graph timed over rotating operand sets; AQL_TILE=<id> forces a tile (one process per tile: the hook is
read once).  usage: python tools/tune_grouped.py   -> one line per shape: us under the picker's choice (or the forced tile)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aqualora_amd import _lib as L

dev, r = "cuda", 320
rnd = lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16)  # noqa: E731
ws = torch.empty(64 << 20, dtype=torch.float32, device=dev)
NSET = 4


def gt(fns, iters=5):
    for f in fns: f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            for f in fns: f()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / (iters * len(fns)))
    return best


def fwd_main(M, C, K, G, row0):
    def mk():
        X, W, Ts, Bup = rnd(M, K), rnd(G * C, K), rnd(M, G * r), rnd(G * C, r)
        Y = torch.empty(M, G * C, dtype=torch.bfloat16, device=dev)
        return lambda: L.call("aql_gemm_bf16_grouped", L.ptr(X), K, L.ptr(W), K, M, G * C, K, L.ptr(Ts), G * r, L.ptr(Bup), r, r, C, 0, r,
                              None, None, 0, L.ptr(Y), G * C, None, 0, None, 1, row0, L.ptr(ws), ws.numel() * 4, L.stream_ptr())
    return gt([mk() for _ in range(NSET)])


def down(M, K, G, rps):
    def mk():
        X, A, S = rnd(M, K), rnd(G * r, K), rnd(M // rps, G * r)
        T = torch.empty(M, G * r, dtype=torch.bfloat16, device=dev); Ts = torch.empty_like(T)
        return lambda: L.call("aql_lora_down", L.ptr(X), K, M, K, L.ptr(A), G * r, L.ptr(S), rps, L.ptr(T), L.ptr(Ts), None, None, L.stream_ptr())
    return gt([mk() for _ in range(NSET)])


def bwd_dts(M, C, G, rps):
    def mk():
        D, Bt, S = rnd(M, G * C), rnd(G * r, C), rnd(M // rps, G * r)
        dTs = torch.empty(M, G * r, dtype=torch.bfloat16, device=dev); dT = torch.empty_like(dTs)
        return lambda: L.call("aql_gemm_bf16_grouped", L.ptr(D), G * C, L.ptr(Bt), C, M, G * r, C, None, 0, None, 0, 0, r, C, 0, None, None, 0,
                              L.ptr(dTs), G * r, L.ptr(dT), G * r, L.ptr(S), rps, 0, None, 0, L.stream_ptr())
    return gt([mk() for _ in range(NSET)])


def bwd_dx(M, C, K, G):
    def mk():
        D, Wt, dT, At = rnd(M, G * C), rnd(K, G * C), rnd(M, G * r), rnd(K, G * r)
        dX = torch.empty(M, K, dtype=torch.bfloat16, device=dev)
        return lambda: L.call("aql_gemm_bf16_ex", L.ptr(D), G * C, L.ptr(Wt), G * C, M, K, G * C, L.ptr(dT), G * r, L.ptr(At), G * r, G * r, None, None, 1,
                              None, 0, L.ptr(dX), K, 0, L.ptr(ws), ws.numel() * 4, L.stream_ptr())
    return gt([mk() for _ in range(NSET)])


tile = os.environ.get("AQL_TILE", "picker")
B = 8
for (C, N) in ((640, 1024), (1280, 256), (1280, 64)):
    Mt, Mh = 2 * B * N, B * N
    print(f"tile={tile} q|k|v C={C} tokens={N}: fwd down {down(Mh, C, 3, N):6.1f}  fwd main {fwd_main(Mt, C, C, 3, Mh):6.1f}  "
          f"bwd dTs {bwd_dts(Mh, C, 3, N):6.1f}  bwd dX {bwd_dx(Mh, C, C, 3):6.1f} us", flush=True)
print(f"tile={tile} q|k|v C=320 chain stages (64x64): bwd dTs {bwd_dts(B * 4096, 320, 3, 4096):6.1f}  bwd dX {bwd_dx(B * 4096, 320, 320, 3):6.1f} us", flush=True)
for C in (320, 640, 1280):
    Mt, Mh = 2 * B * 77, B * 77
    print(f"tile={tile} text k|v C={C}: fwd down {down(Mh, 768, 2, 77):6.1f}  fwd main {fwd_main(Mt, C, 768, 2, Mh):6.1f}  bwd dTs {bwd_dts(Mh, C, 2, 77):6.1f} us", flush=True)
