"""GPU probe: times the bf16 GEMM / conv C-ABI entry points on the U-Net's own shapes (B=4 per pass).  Run it under
different AQL_TILE / AQL_PD values to tune pick_tile():  for t in 5 6 7 8; do AQL_TILE=$t python tools/tune_gemm.py; done"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aqualora_amd import _lib as L

dev = "cuda"
torch.manual_seed(0)
ws = torch.empty(64 << 20, dtype=torch.float32, device=dev)
rnd = lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16)

NSET = int(os.environ.get("NSET", "1"))  # >1: rotate over NSET operand sets (cold L2 / Infinity Cache, like the train step)

def timeit(fns, n=32):
    """fns: one launcher per operand set.  n launches are captured in a HIP graph (no CPU launch gaps), replayed 5x."""
    for f in fns[:2]: f()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for i in range(n): fns[i % len(fns)]()
    g.replay(); torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(5): g.replay()
    t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1) / (5 * n) * 1e3

GEMMS = [(16384, 320, 320), (16384, 2560, 320), (16384, 320, 1280), (4096, 640, 640), (4096, 5120, 640), (4096, 640, 2560),
         (1024, 1280, 1280), (1024, 10240, 1280), (1024, 1280, 5120), (256, 1280, 1280), (308, 640, 768), (308, 1280, 768),
         (16384, 32, 320), (4096, 320, 32)]
tag = f"TILE={os.environ.get('AQL_TILE', '-')} PD={os.environ.get('AQL_PD', '-')} NSET={NSET}"
for (M, N, K) in GEMMS:
    def mk():
        A = rnd(M, K); B = rnd(N, K); C = torch.empty(M, N, dtype=torch.bfloat16, device=dev); bias = rnd(N); R = rnd(M, N)
        return lambda: L.call("aql_gemm_bf16", L.ptr(A), K, L.ptr(B), K, M, N, K, None, 0, None, 0, 0, L.ptr(bias), None, 1,
                              L.ptr(R), N, L.ptr(C), N, L.ptr(ws), ws.numel() * 4, L.stream_ptr())
    us = timeit([mk() for _ in range(NSET)])
    print(f"[{tag}] gemm M{M} N{N} K{K}: {us:7.1f} us {2*M*N*K/us/1e6:7.1f} TF/s", flush=True)
CONVS = [(4, 64, 320, 320), (4, 64, 640, 320), (4, 64, 960, 320), (4, 32, 640, 640), (4, 32, 1280, 640), (4, 32, 1920, 640),
         (4, 16, 1280, 1280), (4, 16, 2560, 1280), (4, 8, 1280, 1280), (4, 8, 2560, 1280)]
for (Bn, H, Cin, Cout) in CONVS:
    def mk():
        xh = rnd(Bn, H, H, Cin); wk = rnd(Cout, 9 * Cin); b = rnd(Cout); y = torch.empty(Bn, H, H, Cout, dtype=torch.bfloat16, device=dev)
        return lambda: L.call("aql_conv3x3_fwd", L.ptr(xh), Bn, H, H, Cin, L.ptr(wk), L.ptr(b), Cout, 1, 0, None, 0, None, L.ptr(y),
                              L.ptr(ws), ws.numel() * 4, L.stream_ptr())
    us = timeit([mk() for _ in range(NSET)])
    print(f"[{tag}] conv B{Bn} {H}x{H} {Cin}->{Cout}: {us:7.1f} us {2*Bn*H*H*Cout*9*Cin/us/1e6:7.1f} TF/s", flush=True)
