"""GEGLU epilogue (aql_gemm_bf16_geglu / aql_lora_gemm_fused_geglu) vs the unfused pair (GEMM, then aql_geglu_fwd): the
activation is applied to the bf16-rounded tile, so G and H must be BIT-IDENTICAL to the two-kernel path; also checked
against fp32 torch, and timed inside a HIP graph.  Prints PASS/FAIL per case and a final verdict."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as Fn  # noqa: E402
from aqualora_amd import _lib as L, ops  # noqa: E402

dev = "cuda"
ok_all = True
rnd = lambda *s, std=1.0: (torch.randn(*s, device=dev) * std).to(torch.bfloat16)  # noqa: E731


class Site:
    def __init__(self, r, K, N):
        self.rank = r
        self.a16, self.b16 = rnd(r, K, std=K ** -0.5), rnd(N, r, std=0.1)
        self.at16, self.bt16 = self.a16.t().contiguous(), self.b16.t().contiguous()


def graph_time(fn, n=20):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(5):
        g.replay()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / (5 * n) * 1e3


def case(M, F, K, rank, nb=4):
    global ok_all
    x = rnd(M, K)
    pk = ops.PackedLinear(torch.randn(2 * F, K, device=dev) * K ** -0.5, torch.randn(2 * F, device=dev) * 0.1)
    site = S16 = None
    if rank:
        site = Site(rank, K, 2 * F)
        S16 = (1.0 + 0.3 * torch.randn(nb, rank, device=dev)).to(torch.bfloat16)
    rps = (M + nb - 1) // nb
    with torch.no_grad():
        os.environ["AQL_GEGLU_FUSED"] = "1"
        yf = ops.LoraLinearFn.apply(x, pk, site, S16, S16, rps, None, True, True)
        os.environ["AQL_GEGLU_FUSED"] = "0"
        yu = ops.LoraLinearFn.apply(x, pk, site, S16, S16, rps, None, True, True)
        os.environ["AQL_GEGLU_FUSED"] = "1"
        # fp32 reference
        h = x.float() @ pk.w.float().t() + pk.bias.float()
        if rank:
            T = (x.float() @ site.a16.float().t()).to(torch.bfloat16).float()
            Ts = (T * S16.float().repeat_interleave(rps, 0)[:M]).to(torch.bfloat16).float()
            h = h + Ts @ site.b16.float().t()
        ref = h[:, :F] * Fn.gelu(h[:, F:])
    same = torch.equal(yf, yu)
    err = ((yf.float() - ref).abs().max() / ref.abs().max()).item()
    good = same and err < 1.5e-2
    ok_all &= good
    tf = graph_time(lambda: ops.LoraLinearFn.apply(x, pk, site, S16, S16, rps, None, True, True))
    tfn = graph_time(lambda: ops.LoraLinearFn.apply(x, pk, site, S16, S16, rps, None, True, False))
    os.environ["AQL_GEGLU_FUSED"] = "0"
    tu = graph_time(lambda: ops.LoraLinearFn.apply(x, pk, site, S16, S16, rps, None, True, True))
    os.environ["AQL_GEGLU_FUSED"] = "1"
    print(f"{'PASS' if good else 'FAIL'} M={M} F={F} K={K} rank={rank}: fused==unfused {same}, vs fp32 {err:.2e} (tol 1.5e-2); "
          f"fused {tf:.1f} us (no H: {tfn:.1f}) vs unfused {tu:.1f} us", flush=True)


def grad_case(M, F, K, rank, nb=2):
    """autograd through the fused op == autograd through the unfused ops (same kernels in backward: equal bits)"""
    global ok_all
    pk = ops.PackedLinear(torch.randn(2 * F, K, device=dev) * K ** -0.5, torch.randn(2 * F, device=dev) * 0.1)
    site = Site(rank, K, 2 * F)
    site.ga = torch.zeros(rank, K, device=dev)
    site.gb = torch.zeros(2 * F, rank, device=dev)
    S = (1.0 + 0.3 * torch.randn(nb, rank, device=dev)).requires_grad_(True)
    S16 = S.detach().to(torch.bfloat16)
    x0 = rnd(M, K)
    dy = rnd(M, F)
    outs = []
    for flag in ("1", "0"):
        os.environ["AQL_GEGLU_FUSED"] = flag
        site.ga.zero_(); site.gb.zero_()
        x = x0.clone().requires_grad_(True)
        S.grad = None
        y = ops.lora_linear(x, pk, site, S, S16, M // nb, None, True)
        y.backward(dy)
        outs.append((y.detach().clone(), x.grad.clone(), S.grad.clone(), site.ga.clone(), site.gb.clone()))
    os.environ["AQL_GEGLU_FUSED"] = "1"
    eq = [torch.equal(a, b) for a, b in zip(outs[0][:2], outs[1][:2])]
    # dS and the weight gradients use fp32 atomics (order-dependent last bits)
    rel = [((a - b).abs().max() / b.abs().max()).item() for a, b in zip(outs[0][2:], outs[1][2:])]
    good = all(eq) and all(r <= 1e-4 for r in rel)
    ok_all &= good
    print(f"{'PASS' if good else 'FAIL'} grad M={M} F={F} K={K} rank={rank}: y, dx equal bits {eq}; dS, dA, dB rel {rel}", flush=True)


for M, F, K in [(16384, 1280, 320), (4096, 2560, 640), (1024, 5120, 1280), (256, 5120, 1280), (1000, 1280, 320), (512, 160, 64),
                (128, 80, 32)]:
    for rank in (0, 32, 8):
        case(M, F, K, rank)
grad_case(2048, 1280, 320, 32)
grad_case(512, 160, 64, 8)


def ff_case(M, C, rank, nb=2):
    """ops.FeedForwardFn: the GEGLU backward in the epilogue of the ff.net.2 backward-data launch == the stand-alone kernel"""
    global ok_all
    F = 4 * C
    pk0 = ops.PackedLinear(torch.randn(2 * F, C, device=dev) * C ** -0.5, torch.randn(2 * F, device=dev) * 0.1)
    pk2 = ops.PackedLinear(torch.randn(C, F, device=dev) * F ** -0.5, torch.randn(C, device=dev) * 0.1)
    s0, s2 = Site(rank, C, 2 * F), Site(rank, F, C)
    for st, K, N in ((s0, C, 2 * F), (s2, F, C)):
        st.ga = torch.zeros(rank, K, device=dev)
        st.gb = torch.zeros(N, rank, device=dev)
    S = (1.0 + 0.3 * torch.randn(nb, rank, device=dev)).requires_grad_(True)
    S16 = S.detach().to(torch.bfloat16)
    x0, res, dy = rnd(M, C), rnd(M, C), rnd(M, C)
    outs = []
    for flag in ("1", "0"):
        os.environ["AQL_GEGLU_BWD_FUSED"] = flag
        for st in (s0, s2):
            st.ga.zero_(); st.gb.zero_()
        x = x0.clone().requires_grad_(True)
        S.grad = None
        y = ops.feed_forward(x, pk0, s0, pk2, s2, S, S16, M // nb, res)
        y.backward(dy)
        outs.append((y.detach().clone(), x.grad.clone(), S.grad.clone(), s0.ga.clone(), s0.gb.clone(), s2.ga.clone(), s2.gb.clone()))
    os.environ["AQL_GEGLU_BWD_FUSED"] = "1"
    eq = [torch.equal(a, b) for a, b in zip(outs[0][:2], outs[1][:2])]
    rel = [((a - b).abs().max() / b.abs().max()).item() for a, b in zip(outs[0][2:], outs[1][2:])]
    # fp32 reference of the whole feed-forward gradient w.r.t. x
    xr = x0.float().requires_grad_(True)
    rows = torch.arange(M, device=dev) // (M // nb)
    def lin(xx, pk, st):
        T = (xx @ st.a16.float().t())
        return xx @ pk.w.float().t() + pk.bias.float() + (T * S16.float()[rows]) @ st.b16.float().t()
    h = lin(xr, pk0, s0)
    yr = lin(h[:, :F] * Fn.gelu(h[:, F:]), pk2, s2) + res.float()
    yr.backward(dy.float())
    e = ((outs[0][1].float() - xr.grad).abs().max() / xr.grad.abs().max()).item()
    good = all(eq) and all(r <= 1e-4 for r in rel) and e < 3e-2
    ok_all &= good
    print(f"{'PASS' if good else 'FAIL'} feed-forward M={M} C={C} rank={rank}: y, dx equal bits {eq}; dS/dA/dB rel {max(rel):.1e}; dx vs fp32 {e:.2e}", flush=True)


ff_case(4096, 320, 32)
ff_case(1024, 640, 32)
ff_case(512, 160, 8)
print("ALL PASS" if ok_all else "SOME FAILED")
