"""Upper bound for a K-concatenated q|k|v backward-data launch: three chained rank-32 LoRA launches (K=320 each, the later
ones adding the earlier result as residual) against ONE launch with K=960 (same FLOPs, one side product instead of three)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from aqualora_amd import ops
dev = "cuda"
rnd = lambda *s, std=1.0: (torch.randn(*s, device=dev) * std).to(torch.bfloat16)
def graph_time(fn, n=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(5): g.replay()
    t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1) / (5 * n) * 1e3
for (M, C, nb) in [(16384, 320, 4), (4096, 640, 4), (1024, 1280, 4)]:
    S16 = (1 + 0.3 * torch.randn(nb, 32, device=dev)).to(torch.bfloat16)
    rps = M // nb
    dys = [rnd(M, C) for _ in range(3)]
    W = [rnd(C, C, std=C ** -0.5) for _ in range(3)]
    A = [rnd(32, C, std=C ** -0.5) for _ in range(3)]
    Bu = [rnd(C, 32, std=0.1) for _ in range(3)]
    T = torch.empty(M, 32, dtype=torch.bfloat16, device=dev); Ts = torch.empty_like(T)
    def three():
        dx = None
        for i in range(3):
            dx = ops._lora_gemm_fused(dys[i], W[i], A[i], S16, rps, Bu[i], None, dx, T, Ts)
        return dx
    dycat = torch.cat(dys, 1).contiguous(); Wcat = torch.cat(W, 1).contiguous(); Acat = torch.cat(A, 1).contiguous()
    def one():
        return ops._lora_gemm_fused(dycat, Wcat, Acat, S16, rps, Bu[0], None, None, T, Ts)
    print(f"M={M} C={C}: three chained launches {graph_time(three):.1f} us, one K={3 * C} launch {graph_time(one):.1f} us")

# the K-grouped kernel itself (aql_lora_gemm_fused_kgroups)
import ctypes
from aqualora_amd import _lib as L
for (M, C, nb) in [(16384, 320, 4), (4096, 640, 4), (1024, 1280, 4)]:
    S16 = (1 + 0.3 * torch.randn(nb, 32, device=dev)).to(torch.bfloat16)
    rps = M // nb
    dys = [rnd(M, C) for _ in range(3)]
    W = [rnd(C, C, std=C ** -0.5) for _ in range(3)]
    A = [rnd(32, C, std=C ** -0.5) for _ in range(3)]
    Bu = [rnd(C, 32, std=0.1) for _ in range(3)]
    dTs = torch.empty(3, M, 32, dtype=torch.bfloat16, device=dev); dT = torch.empty_like(dTs)
    dx = torch.empty(M, C, dtype=torch.bfloat16, device=dev)
    vp, lp_, ip = ctypes.c_void_p * 3, ctypes.c_long * 3, ctypes.c_int * 3
    def kg():
        rc = L.call_raw("aql_lora_gemm_fused_kgroups", 3, vp(*[t.data_ptr() for t in dys]), lp_(C, C, C), vp(*[t.data_ptr() for t in W]),
                        lp_(C, C, C), ip(C, C, C), vp(*[t.data_ptr() for t in A]), vp(*[t.data_ptr() for t in Bu]), M, C, L.ptr(S16), rps,
                        None, 0, L.ptr(dx), C, vp(*[dTs[g].data_ptr() for g in range(3)]), vp(*[dT[g].data_ptr() for g in range(3)]),
                        L.stream_ptr())
        assert rc == 0, rc
    T = torch.empty(M, 32, dtype=torch.bfloat16, device=dev); Ts = torch.empty_like(T)
    def three():
        d = None
        for i in range(3):
            d = ops._lora_gemm_fused(dys[i], W[i], A[i], S16, rps, Bu[i], None, d, T, Ts)
        return d
    kg(); ref = three()
    err = ((dx.float() - ref.float()).abs().max() / ref.float().abs().max()).item()
    print(f"M={M} C={C}: K-grouped launch {graph_time(kg):.1f} us vs three chained {graph_time(three):.1f} us (max rel diff {err:.2e})")
