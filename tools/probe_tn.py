import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aqualora_amd import ops, _lib as L
dev="cuda"
def timeit(fn, n=50):
    fn(); torch.cuda.synchronize()
    t0=torch.cuda.Event(enable_timing=True); t1=torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(n): fn()
    t1.record(); torch.cuda.synchronize(); return t0.elapsed_time(t1)/n*1e3
for (M,P,Q) in [(16384,320,32),(16384,32,320),(16384,2560,32),(4096,640,32),(1024,1280,32),(16384,32,1280)]:
    U=torch.randn(M,P,device=dev).bfloat16(); V=torch.randn(M,Q,device=dev).bfloat16(); C=torch.zeros(P,Q,device=dev)
    res=[]
    for sp in [1,2,4,8,16,32,64,128]:
        os.environ["AQL_TN_SPLITS"]=str(sp)
        res.append("%d:%.1f"%(sp,timeit(lambda: ops.gemm_tn_acc(U,V,C))))
    del os.environ["AQL_TN_SPLITS"]
    print(f"TN M{M} P{P} Q{Q}  default {timeit(lambda: ops.gemm_tn_acc(U,V,C)):.1f} us | "+" ".join(res), flush=True)
