"""One GEMM / conv shape, a handful of launches: the workload for rocprofv3 --pmc passes (tools/pmc_run.sh).
usage: pmc_one.py gemm M N K | conv B H Cin Cout"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aqualora_amd import _lib as L
dev = "cuda"
torch.manual_seed(0)
ws = torch.empty(64 << 20, dtype=torch.float32, device=dev)
rnd = lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16)
kind = sys.argv[1]; a = [int(v) for v in sys.argv[2:]]
NSET = 6
if kind == "gemm":
    M, N, K = a
    def mk():
        A = rnd(M, K); B = rnd(N, K); C = torch.empty(M, N, dtype=torch.bfloat16, device=dev); bias = rnd(N)
        return lambda: L.call("aql_gemm_bf16", L.ptr(A), K, L.ptr(B), K, M, N, K, None, 0, None, 0, 0, L.ptr(bias), None, 1,
                              None, 0, L.ptr(C), N, L.ptr(ws), ws.numel() * 4, L.stream_ptr())
else:
    Bn, H, Cin, Cout = a
    def mk():
        xh = rnd(Bn, H, H, Cin); wk = rnd(Cout, 9 * Cin); b = rnd(Cout); y = torch.empty(Bn, H, H, Cout, dtype=torch.bfloat16, device=dev)
        return lambda: L.call("aql_conv3x3_fwd", L.ptr(xh), Bn, H, H, Cin, L.ptr(wk), L.ptr(b), Cout, 1, 0, None, 0, None, L.ptr(y),
                              L.ptr(ws), ws.numel() * 4, L.stream_ptr())
fns = [mk() for _ in range(NSET)]
for f in fns: f()
torch.cuda.synchronize()
