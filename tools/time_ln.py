"""LayerNorm forward / backward alone on the U-Net's row counts (twin forward 2B, backward B) inside a HIP graph; AQL_LN_ROWS=1|2."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aqualora_amd import _lib as L
for M, C in ((32768, 320), (16384, 320), (8192, 640), (4096, 640), (2048, 1280), (1024, 1280), (512, 1280)):
    x = torch.randn(M, C, device="cuda").to(torch.bfloat16); dy = torch.randn_like(x); res = torch.randn_like(x)
    g = torch.ones(C, device="cuda", dtype=torch.bfloat16); b = torch.zeros_like(g)
    y = torch.empty_like(x); st = torch.empty(M, 2, device="cuda"); dx = torch.empty_like(x)
    fw = lambda: L.call("aql_layernorm_fwd", L.ptr(x), M, C, L.ptr(g), L.ptr(b), 1e-5, L.ptr(y), L.ptr(st), L.stream_ptr())
    bw = lambda: L.call("aql_layernorm_bwd", L.ptr(x), L.ptr(dy), M, C, L.ptr(g), L.ptr(st), L.ptr(res), L.ptr(dx), L.stream_ptr())
    out = []
    for f in (fw, bw):
        f(); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(20): f()
        gr.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); [gr.replay() for _ in range(5)]; e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3 / 100)
    mb = M * C * 2 / 1e6
    print(f"M{M:6d} C{C:5d}: fwd {out[0]:5.1f} us ({2 * mb / out[0] / 1e3:.2f} TB/s)  bwd {out[1]:5.1f} us ({4 * mb / out[1] / 1e3:.2f} TB/s)", flush=True)
