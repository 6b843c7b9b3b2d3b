"""Round 6: the finalize launch + GroupNorm launch against the one-launch slab form, forward and backward, on the U-Net's split-K maps
(HIP graph of 20 launch sets, microseconds per set).  The slabs are filled once; both forms read the same workspace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aqualora_amd import _lib as L

dev = "cuda"


def bench(f):
    f(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(20): f()
    gr.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); [gr.replay() for _ in range(5)]; e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / 100)
    return best


for B, H, C, splits in ((2, 8, 1280, 8), (2, 16, 1280, 8), (2, 32, 640, 4), (8, 8, 1280, 8), (8, 16, 1280, 2), (8, 16, 1280, 4), (8, 32, 640, 2),
                        (4, 8, 1280, 8), (4, 16, 1280, 4), (4, 16, 640, 4), (4, 32, 1920, 2), (16, 16, 1280, 2), (16, 8, 1280, 4)):
    HW, M = H * H, B * H * H
    ws = torch.randn(splits * M * C, device=dev)
    bias = torch.randn(C, device=dev).to(torch.bfloat16); rb = torch.randn(B, C, device=dev).to(torch.bfloat16)
    g = torch.ones(C, device=dev, dtype=torch.bfloat16); be = torch.zeros_like(g)
    x = torch.empty(M, C, device=dev, dtype=torch.bfloat16); y = torch.empty_like(x); dx = torch.empty_like(x); dn = torch.empty_like(x)
    st = torch.empty(B, 32, 2, device=dev); scr = torch.empty(1 << 20, device=dev)
    sp = None

    def two_f():
        L.call("aql_splitk_finalize", L.ptr(ws), splits, M, C, L.ptr(bias), L.ptr(rb), C, HW, None, 0, L.ptr(x), C, L.stream_ptr())
        L.call("aql_groupnorm_silu_fwd", L.ptr(x), B, HW, C, L.ptr(g), L.ptr(be), 1e-5, 1, L.ptr(y), L.ptr(st), L.ptr(scr), L.stream_ptr())

    def one_f():
        return L.call_raw("aql_groupnorm_silu_fwd_slabs", L.ptr(ws), splits, L.ptr(bias), L.ptr(rb), C, None, L.ptr(x), B, HW, C, L.ptr(g), L.ptr(be),
                          1e-5, 1, L.ptr(y), L.ptr(st), L.stream_ptr())

    def gn_f():
        L.call("aql_groupnorm_silu_fwd", L.ptr(x), B, HW, C, L.ptr(g), L.ptr(be), 1e-5, 1, L.ptr(y), L.ptr(st), L.ptr(scr), L.stream_ptr())

    def two_b():
        L.call("aql_splitk_finalize", L.ptr(ws), splits, M, C, None, None, C, 1, None, 0, L.ptr(dn), C, L.stream_ptr())
        L.call("aql_groupnorm_silu_bwd", L.ptr(x), L.ptr(dn), B, HW, C, L.ptr(g), L.ptr(be), 1, L.ptr(st), None, L.ptr(dx), L.ptr(scr), L.stream_ptr())

    def one_b():
        return L.call_raw("aql_groupnorm_silu_bwd_slabs", L.ptr(x), L.ptr(ws), splits, B, HW, C, L.ptr(g), L.ptr(be), 1, L.ptr(st), None, L.ptr(dx), L.stream_ptr())

    two_f()
    line = f"B={B:2d} {H:2d}x{H:<2d} C={C:4d} splits={splits}:"
    t2, tg = bench(two_f), bench(gn_f)
    t1 = bench(one_f) if one_f() != 100 else float("nan")
    line += f"  fwd finalize+GN {t2:5.1f} (GN alone {tg:5.1f})  slabs {t1:5.1f} us"
    t2 = bench(two_b)
    t1 = bench(one_b) if one_b() != 100 else float("nan")
    line += f"  | bwd finalize+GN {t2:5.1f}  slabs {t1:5.1f} us"
    print(line, flush=True)
