"""Per-workgroup phase timeline of the 4-wave one-launch LoRA linear (library built with -DAQL_TRACE_L, selected with
AQL_LIB): where a workgroup's life goes on the short-K shapes.  Usage:
    AQL_LIB=.../libaqualora_trace.so python tools/trace_lora.py
Phases (cycles of s_memtime, wave 0): ring fill issue, first tile landed, K loop, LoRA up step, epilogue tile->LDS, global
stores issued, stores retired.  Also prints how many workgroups were co-resident per CU."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from aqualora_amd import _lib as L, ops  # noqa: E402

dev = "cuda"
rnd = lambda *s, std=1.0: (torch.randn(*s, device=dev) * std).to(torch.bfloat16)  # noqa: E731


class Site:
    def __init__(self, r, K, N):
        self.rank = r
        self.a16, self.b16 = rnd(r, K, std=K ** -0.5), rnd(N, r, std=0.1)
        self.at16, self.bt16 = self.a16.t().contiguous(), self.b16.t().contiguous()


def run(M, N, K, geglu, nb=8, twin=True, cfg=None):
    pk = ops.PackedLinear(torch.randn(N, K, device=dev) * K ** -0.5, torch.randn(N, device=dev) * 0.1)
    site = Site(32, K, N)
    rps = M // nb
    if twin:   # M rows = clean half + watermarked half, as in the train step
        ops.dual_begin()
        x = ops.make_twin(rnd(M // 2, K), rnd(M // 2, K))
        S16 = ops.make_twin(torch.zeros(nb // 2, 32, device=dev, dtype=torch.bfloat16),
                            (1.0 + 0.3 * torch.randn(nb // 2, 32, device=dev)).to(torch.bfloat16))
    else:
        x = rnd(M, K)
        S16 = (1.0 + 0.3 * torch.randn(nb, 32, device=dev)).to(torch.bfloat16)
    buf = torch.zeros(1 << 20, dtype=torch.int64, device=dev)
    os.environ["AQL_TRACE_BUF"] = hex(buf.data_ptr())
    if cfg:
        os.environ["AQL_LORA_CFG"] = cfg
    else:
        os.environ.pop("AQL_LORA_CFG", None)

    def call():
        with torch.no_grad():
            return ops.LoraLinearFn.apply(x, pk, site, S16, S16, rps, None, geglu, True)

    # time without the trace side effects mattering (they are a handful of stores per workgroup)
    for _ in range(3):
        y = call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        call()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    buf.zero_()
    call()
    torch.cuda.synchronize()
    t = buf.view(-1, 16).cpu()
    live = t[:, 0] != 0
    t = t[live]
    nblk = t.shape[0]
    if nblk == 0:
        print(f"M={M} N={N} K={K} geglu={geglu}: {us:.1f} us -- no trace (not the 4-wave kernel?)")
        return
    t0 = t[:, 0].double()
    ph = [(t[:, i + 1] - t[:, i]).double() for i in range(7)]
    names = ["issue ring", "first tile", "K loop", "up step", "tile->LDS", "stores", "retire"]
    hw, xcc = t[:, 8], t[:, 9] & 0xF
    cu = ((hw >> 8) & 0xF) | (((hw >> 13) & 0x7) << 4) | (((hw >> 12) & 1) << 7)
    key = xcc * 256 + cu
    start, end = t0, t[:, 7].double()
    span = float(end.max() - start.min())
    # co-residency: for every workgroup, how many others on the same CU overlap its midpoint
    mid = (start + end) / 2
    co = []
    for k in key.unique():
        sel = key == k
        s_, e_, m_ = start[sel], end[sel], mid[sel]
        co.append(((s_[None, :] <= m_[:, None]) & (e_[None, :] >= m_[:, None])).sum(1).double().mean().item())
    print(f"M={M} N={N} K={K} geglu={geglu} cfg={cfg}: {us:.1f} us, {nblk} workgroups on {len(key.unique())} CUs, kernel span {span:.0f} cyc "
          f"({span / us / 1e3:.2f} GHz), life {float((end - start).mean()):.0f} cyc, co-resident {sum(co) / len(co):.2f}")
    print("   " + "  ".join(f"{n} {float(p.mean()):.0f}" for n, p in zip(names, ph)))
    if int(t[0, 10]) != 0:
        print(f"   issue ring = setup {float((t[:, 10] - t[:, 0]).double().mean()):.0f} + register prefetches "
              f"{float((t[:, 11] - t[:, 10]).double().mean()):.0f} + DMA issue {float((t[:, 1] - t[:, 11]).double().mean()):.0f}")
    # phase relation of co-resident workgroups: fraction of a workgroup's K loop [t2, t3] that lies inside the K loop of another
    # workgroup of the same CU (1.0 = the two share the matrix pipe for their whole K loops, 0 = perfectly de-phased)
    k0, k1 = t[:, 2].double(), t[:, 3].double()
    ov = []
    for k in key.unique():
        sel = (key == k).nonzero().flatten()
        a0, a1 = k0[sel], k1[sel]
        inter = (torch.minimum(a1[:, None], a1[None, :]) - torch.maximum(a0[:, None], a0[None, :])).clamp(min=0)
        inter.fill_diagonal_(0)
        ov.append((inter.sum(1) / (a1 - a0).clamp(min=1)).mean().item())
    slot = (hw & 0xF)
    print(f"   K-loop overlap with co-resident workgroups {sum(ov) / len(ov):.2f}; wave-slot histogram of wave 0 "
          f"{torch.bincount(slot.long(), minlength=8).tolist()}")
    if os.environ.get("TRACE_DUMP"):
        k = key.unique()[0]
        sel = (key == k).nonzero().flatten()
        base = float(start.min())
        for i in sel[start[sel].argsort()][:24]:
            print(f"      cu0 wg: slot {int(slot[i])} simd {int((hw[i] >> 4) & 3)} start {float(start[i]) - base:8.0f} kloop "
                  f"{float(k0[i]) - base:8.0f}..{float(k1[i]) - base:8.0f} end {float(end[i]) - base:8.0f}")
    if twin:
        ops.dual_end()


if __name__ == "__main__":
    torch.manual_seed(0)
    run(32768, 2560, 320, True)
    run(32768, 2560, 320, True, cfg="d64s")
    run(16384, 2560, 320, True, twin=False, nb=4)
    run(32768, 960, 320, False)
    run(32768, 320, 320, False)
    run(32768, 320, 1280, False)
    run(8192, 640, 640, False)
    run(8192, 5120, 640, True)
