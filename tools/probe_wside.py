"""Weight-side LoRA form (ops.wside_*: per-sample effective weights, aql_gemm_bf16_sw) against the activation-side path on the same
inputs: forward outputs, dX, dA, dBup, dS for a square site, ff.net.0 + GEGLU, ff.net.2 with the GEGLU-backward epilogue (through
FeedForwardFn), plain and twin batches; and HIP-graph timings of forward + backward of both forms.  PASS/FAIL lines."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from aqualora_amd import lora as AL, ops  # noqa: E402

dev = "cuda"
torch.manual_seed(0)
ok_all = True


def rel(a, b):
    return float((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-30))


def make_site(cin, cout, r):
    host = AL.LoRACompatibleLinear(cin, cout, device=dev, dtype=torch.bfloat16)
    ll = AL.LoRALinearLayer(cin, cout, r, device=dev, dtype=torch.float32)
    with torch.no_grad():
        host.weight.copy_(torch.randn(cout, cin, device=dev) * cin ** -0.5)
        host.bias.copy_(torch.randn(cout, device=dev) * 0.05)
        ll.down.weight.copy_(torch.randn(r, cin, device=dev) / r)
        ll.up.weight.copy_(torch.randn(cout, r, device=dev) * 0.05)
    host.set_lora_layer(ll)
    return host, AL._packed_linear(host), AL._site_of(ll)


def grads(site):
    ga, gb = site.ga.clone(), site.gb.clone()
    site.ga.zero_()
    site.gb.zero_()
    return ga, gb


def run_linear(wside, x, packed, site, S, rps, res):
    ops._WSIDE = wside
    xg = x.clone().requires_grad_(True)
    Sg = S.clone().requires_grad_(True)
    S16 = S.detach().to(torch.bfloat16)
    y = ops.LoraLinearFn.apply(xg, packed, site, Sg, S16, rps, res, False, True)
    w = torch.randn_like(y, dtype=torch.float32).to(torch.bfloat16)
    torch.manual_seed(1)
    (y.float() * torch.randn_like(y, dtype=torch.float32)).sum().backward()
    return y.detach(), xg.grad, Sg.grad, grads(site)


def run_ff(wside, x, p0, s0, p2, s2, S, rps, res):
    ops._WSIDE = wside
    xg = x.clone().requires_grad_(True)
    Sg = S.clone().requires_grad_(True)
    S16 = S.detach().to(torch.bfloat16)
    y = ops.feed_forward(xg, p0, s0, p2, s2, Sg, S16, rps, res)
    torch.manual_seed(1)
    (y.float() * torch.randn_like(y, dtype=torch.float32)).sum().backward()
    return y.detach(), xg.grad, Sg.grad, grads(s0), grads(s2)


def cmp(tag, a, b, tol):
    global ok_all
    errs = []

    def walk(x, y):
        if isinstance(x, tuple):
            for u, v in zip(x, y):
                walk(u, v)
        else:
            errs.append(rel(x, y))
    walk(a, b)
    good = max(errs) < tol and all(e == e for e in errs)
    ok_all &= good
    print(f"{'PASS' if good else 'FAIL'} {tag}: rel L2 weight-side vs activation-side " + " ".join(f"{e:.2e}" for e in errs), flush=True)


for B, rps, C, r in ((2, 1024, 320, 320), (3, 256, 320, 320), (2, 4096, 320, 320)):
    M = B * rps
    x = (torch.randn(M, C, device=dev)).to(torch.bfloat16)
    S = 1.0 + 0.3 * torch.randn(B, r, device=dev)
    res = torch.randn(M, C, device=dev).to(torch.bfloat16)
    host, packed, site = make_site(C, C, r)
    site.refresh(True)
    ops._WSIDE = True
    assert ops.wside_ok(packed, site, S.to(torch.bfloat16), rps) == (rps >= 1024), (rps, ops.wside_ok(packed, site, S.to(torch.bfloat16), rps))
    if rps < 1024:
        continue
    a = run_linear(True, x, packed, site, S, rps, res)
    b = run_linear(False, x, packed, site, S, rps, res)
    cmp(f"square {C}->{C} r{r} B{B} rps{rps} (y, dX, dS, (dA, dBup))", a, b, 3e-2)
    _, p0, s0 = make_site(C, 8 * C, r)
    _, p2, s2 = make_site(4 * C, C, r)
    s0.refresh(True)
    s2.refresh(True)
    a = run_ff(True, x, p0, s0, p2, s2, S, rps, res)
    b = run_ff(False, x, p0, s0, p2, s2, S, rps, res)
    cmp(f"feed-forward {C}->{8 * C}->GEGLU->{C} r{r} B{B} rps{rps} (y, dX, dS, (dA0, dB0), (dA2, dB2))", a, b, 3e-2)


def graph_time(fn, n=5):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        g.replay()
        t1.record()
        torch.cuda.synchronize()
        best = min(best, t0.elapsed_time(t1) / n * 1e3)
    return best


# forward + backward of one site at BASELINE config 3's 64x64 level (8 samples x 4096 tokens, rank 320), both forms
B, rps, C, r = 8, 4096, 320, 320
x = torch.randn(B * rps, C, device=dev).to(torch.bfloat16)
S = 1.0 + 0.3 * torch.randn(B, r, device=dev)
_, packed, site = make_site(C, C, r)
_, p0, s0 = make_site(C, 8 * C, r)
_, p2, s2 = make_site(4 * C, C, r)
for st in (site, s0, s2):
    st.refresh(True)
for tag, fn in (("square 320->320", lambda w: run_linear(w, x, packed, site, S, rps, None)),
                ("feed-forward 320->2560->320", lambda w: run_ff(w, x, p0, s0, p2, s2, S, rps, None))):
    tw = graph_time(lambda: fn(True))
    ta = graph_time(lambda: fn(False))
    print(f"time {tag}, 8 x 4096 tokens, rank 320, forward + backward incl. weight gradients: weight-side {tw:.0f} us, "
          f"activation-side {ta:.0f} us ({ta / tw:.2f}x)", flush=True)
print("ALL PASS" if ok_all else "SOME FAILED")
sys.exit(0 if ok_all else 1)
