"""GroupNorm(+SiLU) forward / backward on the U-Net's shapes (B=4), timed inside a HIP graph of 20 launches so that the
launch overhead of the 1-launch and 3-launch forms does not enter.  AQL_GN_FUSED=0|1|2 selects the form."""
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from aqualora_amd import ops  # noqa: E402
from aqualora_amd import _lib as L  # noqa: E402

SHAPES = [(320, 64), (640, 64), (960, 64), (320, 32), (640, 32), (960, 32), (1280, 32), (1920, 32), (640, 16), (1280, 16),
          (1920, 16), (2560, 16), (1280, 8), (2560, 8)]
B = int(__import__("os").environ.get("GN_B", "4"))
dev = "cuda"


def graph_time(fn, n=20, reps=10):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (n * reps)


for C, H in SHAPES:
    x = torch.randn(B, H, H, C, device=dev).to(torch.bfloat16)
    dy = torch.randn_like(x)
    y = torch.empty_like(x)
    dx = torch.empty_like(x)
    ga = torch.randn(C, device=dev).to(torch.bfloat16)
    be = torch.randn(C, device=dev).to(torch.bfloat16)
    stats = torch.empty(B, 32, 2, device=dev)
    scr = torch.zeros(1 << 18, device=dev)

    def fwd():
        L.call("aql_groupnorm_silu_fwd", L.ptr(x), B, H * H, C, L.ptr(ga), L.ptr(be), 1e-5, 1, L.ptr(y), L.ptr(stats), L.ptr(scr),
               L.stream_ptr())

    def bwd():
        L.call("aql_groupnorm_silu_bwd", L.ptr(x), L.ptr(dy), B, H * H, C, L.ptr(ga), L.ptr(be), 1, L.ptr(stats), None, L.ptr(dx),
               L.ptr(scr), L.stream_ptr())

    tf = graph_time(fwd)
    tb = graph_time(bwd)
    mb = x.numel() * 2 / 1e6
    print(f"C{C:5d} {H:2d}x{H:<2d} slice {H * H * C // 32 * 2 / 1024:6.0f} KB  fwd {tf:6.1f} us ({2 * mb / tf * 1e-3:5.2f} TB/s)  "
          f"bwd {tb:6.1f} us ({3 * mb / tb * 1e-3:5.2f} TB/s)")
