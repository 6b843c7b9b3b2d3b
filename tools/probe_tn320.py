"""TN (weight-gradient) GEMMs at rank 320, batch 8 (BASELINE config 3): splits sweep."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aqualora_amd import ops
dev = "cuda"
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(n): fn()
    t1.record(); torch.cuda.synchronize(); return t0.elapsed_time(t1) / n * 1e3
for (M, P, Q) in [(32768, 320, 320), (32768, 2560, 320), (32768, 320, 1280), (8192, 640, 320), (2048, 1280, 320), (2048, 320, 5120), (616, 320, 768)]:
    U = torch.randn(M, P, device=dev).bfloat16(); V = torch.randn(M, Q, device=dev).bfloat16(); C = torch.zeros(P, Q, device=dev)
    res = []
    for sp in [1, 2, 4, 8, 16, 32, 64]:
        os.environ["AQL_TN_SPLITS"] = str(sp)
        res.append("%d:%.0f" % (sp, timeit(lambda: ops.gemm_tn_acc(U, V, C))))
    del os.environ["AQL_TN_SPLITS"]
    d = timeit(lambda: ops.gemm_tn_acc(U, V, C))
    print(f"TN M{M} P{P} Q{Q}  default {d:.0f} us = {2*M*P*Q/d/1e6:.0f} TF/s | " + " ".join(res), flush=True)
