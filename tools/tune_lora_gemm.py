"""Fused rank-32 LoRA linear (aql_lora_gemm_fused) on the U-Net's shapes inside a HIP graph (cold operand rotation), next
to the two-launch form (aql_lora_down + aql_gemm_bf16).  AQL_LORA_BM=32|64|128 forces the fused kernel's tile height."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from aqualora_amd import _lib as L, ops  # noqa: E402

dev = "cuda"
NSET = 8
SHAPES = [(16384, 320, 320), (16384, 2560, 320), (16384, 320, 1280), (4096, 640, 640), (4096, 5120, 640), (4096, 640, 2560),
          (1024, 1280, 1280), (1024, 10240, 1280), (1024, 1280, 5120), (256, 1280, 1280), (308, 640, 768)]


def graph_time(fns, n=32):
    for f in fns[:2]:
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(n):
            fns[i % len(fns)]()
    g.replay()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(5):
        g.replay()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / (5 * n) * 1e3


rnd = lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16)  # noqa: E731
for M, N, K in SHAPES:
    def mk(fused):
        X, W, A, Bup, S = rnd(M, K), rnd(N, K), rnd(32, K), rnd(N, 32), rnd(4, 32)
        Y, T, Ts = torch.empty(M, N, dtype=torch.bfloat16, device=dev), torch.empty(M, 32, dtype=torch.bfloat16, device=dev), \
            torch.empty(M, 32, dtype=torch.bfloat16, device=dev)
        rps = M // 4
        if fused:
            def f():
                rc = L.call_raw("aql_lora_gemm_fused", L.ptr(X), K, L.ptr(W), K, M, N, K, L.ptr(A), L.ptr(S), rps, L.ptr(Bup), None,
                                None, 0, L.ptr(Y), N, L.ptr(T), L.ptr(Ts), 0, L.stream_ptr())
                assert rc in (0, 100), rc
                return rc
            return f

        def f2():
            L.call("aql_lora_down", L.ptr(X), K, M, K, L.ptr(A), 32, L.ptr(S), rps, L.ptr(T), L.ptr(Ts), None, None, L.stream_ptr())
            ops.gemm_bf16(X, W, None, Ts, Bup, out=Y)
        return f2
    probe = mk(True)
    if probe() == 100:
        print(f"M{M} N{N} K{K}: not fused (two-launch path)")
        continue
    tf = graph_time([mk(True) for _ in range(NSET)])
    t2 = graph_time([mk(False) for _ in range(NSET)])
    fl = 2.0 * M * K * (N + 32) + 2.0 * M * 32 * N
    print(f"M{M:6d} N{N:6d} K{K:5d}: fused {tf:7.1f} us ({fl / tf / 1e6:6.1f} TF/s)   two-launch {t2:7.1f} us", flush=True)
