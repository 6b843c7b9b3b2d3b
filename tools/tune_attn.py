"""Time the attention kernels alone on the U-Net's shapes (B=4): forward, backward (delta + dQ + dK/dV), and check the
forward against torch SDPA.  Usage: python tools/tune_attn.py [iters]"""
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from aqualora_amd import _lib as L  # noqa: E402
import os
L.LIB_PATH = os.environ.get("AQL_LIB", L.LIB_PATH)   # ablation builds

SHAPES = [(4, 8, 4096, 4096, 40), (4, 8, 1024, 1024, 80), (4, 8, 256, 256, 160), (4, 8, 4096, 77, 40),
          (4, 8, 1024, 77, 80), (4, 8, 256, 77, 160), (4, 8, 64, 64, 160)]


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    dev = "cuda"
    torch.manual_seed(0)
    print(f"{'B H Nq Nk d':>22} {'fwd us':>8} {'TF/s':>7} {'bwd us':>8} {'TF/s':>7}  relerr")
    for B, H, Nq, Nk, d in SHAPES:
        C = H * d
        q = torch.randn(B, Nq, C, device=dev, dtype=torch.bfloat16)
        k = torch.randn(B, Nk, C, device=dev, dtype=torch.bfloat16)
        v = torch.randn(B, Nk, C, device=dev, dtype=torch.bfloat16)
        do = torch.randn(B, Nq, C, device=dev, dtype=torch.bfloat16)
        o = torch.empty_like(q)
        lse = torch.empty(B, H, Nq, device=dev, dtype=torch.float32)
        delta = torch.empty_like(lse)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        st = L.stream_ptr()
        ws = torch.empty(16 << 20, device=dev, dtype=torch.float32)
        sc = float(d ** -0.5)

        def fwd():
            L.call("aql_sdpa_fwd", L.ptr(q), q.stride(1), L.ptr(k), k.stride(1), L.ptr(v), v.stride(1), B, H, Nq, Nk, d, sc,
                   L.ptr(o), o.stride(1), L.ptr(lse), st)

        def bwd():
            L.call("aql_sdpa_bwd", L.ptr(q), q.stride(1), L.ptr(k), k.stride(1), L.ptr(v), v.stride(1), L.ptr(o), L.ptr(do),
                   o.stride(1), L.ptr(lse), L.ptr(delta), B, H, Nq, Nk, d, sc, L.ptr(dq), L.ptr(dk), L.ptr(dv), L.ptr(ws),
                   ws.numel() * 4, st)

        tf = timeit(fwd, iters)
        tb = timeit(bwd, iters)
        ref = torch.nn.functional.scaled_dot_product_attention(
            q.view(B, Nq, H, d).transpose(1, 2).float(), k.view(B, Nk, H, d).transpose(1, 2).float(),
            v.view(B, Nk, H, d).transpose(1, 2).float()).transpose(1, 2).reshape(B, Nq, C)
        err = float((o.float() - ref).abs().max() / ref.abs().max())
        qr, kr, vr = (t.float().view(B, -1, H, d).transpose(1, 2).detach().requires_grad_(True) for t in (q, k, v))
        orf = torch.nn.functional.scaled_dot_product_attention(qr, kr, vr)
        orf.backward(do.float().view(B, Nq, H, d).transpose(1, 2))
        for name, got, want in (("dq", dq, qr.grad), ("dk", dk, kr.grad), ("dv", dv, vr.grad)):
            w = want.transpose(1, 2).reshape(got.shape)
            err = max(err, float((got.float() - w).abs().max() / w.abs().max()))
        fl = 4.0 * B * H * Nq * Nk * d
        print(f"{str((B, H, Nq, Nk, d)):>22} {tf:8.1f} {fl / tf / 1e6:7.1f} {tb:8.1f} {2.5 * fl / tb / 1e6:7.1f}  {err:.2e}")


if __name__ == "__main__":
    main()
