"""(QPRE=1: the entries that take a q pre-multiplied by d^-1/2 log2(e), aql_sdpa_fwd_qpre / _bwd_qpre.)
HIP-graph timing of the self-attention kernels alone (forward; with BWD=1 also dQ and dK/dV) on the U-Net's two large
shapes, twin (8 samples) and batch-4.  AQL_LIB selects an ablation build, AQL_ATTN32=0 the 16x16x32 kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aqualora_amd import _lib as L

def graph_time(fn, iters=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters): fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / iters)
    return best

torch.manual_seed(0)
out = []
SHAPES = ((4, 8, 4096, 40), (8, 8, 4096, 40), (4, 8, 1024, 80), (8, 8, 1024, 80))
if os.environ.get("SMALL"):   # the low-resolution levels: 16x16 (d = 160) and 8x8
    SHAPES = ((4, 8, 256, 160), (8, 8, 256, 160), (4, 8, 64, 160), (8, 8, 64, 160), (4, 8, 1024, 80), (8, 8, 1024, 80))
if os.environ.get("SHAPES"):   # SHAPES="2,8,4096,40;2,8,1024,80": B,H,N,d per shape (inference: CFG batch 2)
    SHAPES = tuple(tuple(int(v) for v in sh.split(",")) for sh in os.environ["SHAPES"].split(";"))
NK = int(os.environ.get("CTX", "0"))   # CTX=77: the text-state (cross-attention) shapes, forward at 8 samples, backward at 4
if NK:
    SHAPES = ((8, 8, 4096, 40), (4, 8, 4096, 40), (8, 8, 1024, 80), (4, 8, 1024, 80), (8, 8, 256, 160), (4, 8, 256, 160), (8, 8, 64, 160), (4, 8, 64, 160))
for B, H, N, d in SHAPES:
    C = H * d
    q, do = (torch.randn(B, N, C, device="cuda", dtype=torch.bfloat16) for _ in range(2))
    k, v = (torch.randn(B, NK or N, C, device="cuda", dtype=torch.bfloat16) for _ in range(2))
    o = torch.empty_like(q); lse = torch.empty(B, H, N, device="cuda"); delta = torch.empty_like(lse)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    ws = torch.empty(16 << 20, device="cuda"); sc = float(d ** -0.5); st = L.stream_ptr
    SFX = "_qpre" if os.environ.get("QPRE") == "1" else ""
    fwd = lambda: L.call("aql_sdpa_fwd" + SFX, L.ptr(q), C, L.ptr(k), C, L.ptr(v), C, B, H, N, NK or N, d, sc, L.ptr(o), C, L.ptr(lse), st())
    bwd = lambda: L.call("aql_sdpa_bwd" + SFX, L.ptr(q), C, L.ptr(k), C, L.ptr(v), C, L.ptr(o), L.ptr(do), C, L.ptr(lse), L.ptr(delta), B, H, N, NK or N,
                         d, sc, L.ptr(dq), L.ptr(dk), L.ptr(dv), L.ptr(ws), ws.numel() * 4, st())
    tf = graph_time(fwd)
    fl = 4.0 * B * H * N * (NK or N) * d
    line = f"B={B} N={N} d={d}: fwd {tf:7.1f} us ({fl / tf / 1e6:6.0f} TF/s = {fl / tf / 1e6 / 2500:.3f})"
    if os.environ.get("BWD"):
        tb = graph_time(bwd)
        line += f"  bwd {tb:7.1f} us ({2.5 * fl / tb / 1e6:6.0f} TF/s)"
    print(line, flush=True)
