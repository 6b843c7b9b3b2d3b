"""Weight-gradient (token-reduction) GEMMs of config 3 (rank 320, 8 samples) and config 2 (rank 32, 4 samples): aql_gemm_tn_tr_f32
inside HIP graphs of 20 launches on rotating operands; AQL_TNTR_NST selects the body (0 = register prefetch, 2 / 3 / 4 = DMA ring)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from aqualora_amd import ops
dev = "cuda"
def gt(fns, n=20):
    for f in fns: f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(n): fns[i % len(fns)]()
    g.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1e3)
    return sorted(ts)[2]
tot = 0.0
for (M, P, Q) in [(32768, 320, 320), (32768, 320, 1280), (32768, 2560, 320), (32768, 1280, 320), (8192, 640, 320), (8192, 320, 2560), (2048, 1280, 320),
                  (2048, 320, 5120), (616, 320, 768), (16384, 32, 320), (16384, 320, 32), (16384, 32, 1280), (4096, 640, 32), (1024, 32, 5120)]:
    sets = []
    for _ in range(4):
        U = torch.randn(M, P, device=dev).bfloat16(); V = torch.randn(M, Q, device=dev).bfloat16(); C = torch.zeros(P, Q, device=dev)
        sets.append((U, V, C))
    us = gt([(lambda s=s: ops.gemm_tn_acc(*s)) for s in sets])
    tot += us
    print(f"TN M{M:6d} P{P:5d} Q{Q:5d}: {us:7.1f} us = {2.0 * M * P * Q / us / 1e6:6.0f} TF/s", flush=True)
print(f"sum {tot:.1f} us")
