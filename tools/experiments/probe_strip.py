"""X-stationary strip kernel (aql_gemm_strip.cuh, AQL_STRIP=1) vs the tiled one-launch LoRA linear (AQL_STRIP=0) on the short-K
shapes of the U-Net: outputs (Y or G/H, T, Ts) must be BIT-IDENTICAL; both are timed inside a HIP graph.  Prints PASS/FAIL per
case and a final verdict."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from aqualora_amd import ops  # noqa: E402

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


class Grab:
    """captures what LoraLinearFn saves for backward (T, Ts, H)"""
    def save_for_backward(self, *t):
        self.saved = t
    needs_input_grad = (False,) * 9


def case(M, N, K, geglu=False, res=False, twin=False, nb=4, force=None):
    global ok_all
    pk = ops.PackedLinear(torch.randn(N, K, device=dev) * K ** -0.5, torch.randn(N, device=dev) * 0.1)
    site = Site(32, K, N)
    rps = M // nb
    if twin:
        ops.dual_begin()
        x = ops.make_twin(rnd(M // 2, K), rnd(M // 2, K))
        S16 = ops.make_twin(torch.zeros(nb // 2, 32, device=dev, dtype=torch.bfloat16),
                            (1.0 + 0.3 * torch.randn(nb // 2, 32, device=dev)).to(torch.bfloat16))
        r = ops.make_twin(rnd(M // 2, N), rnd(M // 2, N)) if res else None
    else:
        x = rnd(M, K)
        S16 = (1.0 + 0.3 * torch.randn(nb, 32, device=dev)).to(torch.bfloat16)
        r = rnd(M, N) if res else None

    def run(flag):
        os.environ["AQL_STRIP"] = flag
        ctx = Grab()
        with torch.no_grad():
            y = ops.LoraLinearFn.forward(ctx, x, pk, site, S16, S16, rps, r, geglu, True)
        # the output on BOTH halves of a twin batch; T, Ts and the saved pre-activation only where they are defined (second half)
        yk = ops._full(y) if twin else y
        return [yk] + [t for t in ctx.saved[1:] if t is not None and t is not S16]

    a = run(force or "1")
    b = run("0")
    same = [torch.equal(p, q) for p, q in zip(a, b)]
    ts = graph_time(lambda: run(force or "1"))
    tt = graph_time(lambda: run("0"))
    os.environ.pop("AQL_STRIP", None)
    good = all(same) and len(a) == len(b)
    ok_all &= good
    print(f"{'PASS' if good else 'FAIL'} M={M} N={N} K={K} geglu={geglu} res={res} twin={twin}: equal bits {same}; "
          f"strip {ts:.1f} us vs tiled {tt:.1f} us ({tt / ts:.2f}x)", flush=True)
    if twin:
        ops.dual_end()


if __name__ == "__main__":
    torch.manual_seed(0)
    case(32768, 2560, 320, geglu=True, twin=True, nb=8)     # ff.net.0 at 64x64, twin batch
    case(16384, 2560, 320, geglu=True)
    case(32768, 320, 320, res=True, twin=True, nb=8)        # to_out / attn2.to_q
    case(16384, 320, 320, res=True)
    case(32768, 960, 320, twin=True, nb=8)
    case(8192, 5120, 640, geglu=True, twin=True, nb=8)      # ff.net.0 at 32x32
    case(8192, 640, 640, res=True, twin=True, nb=8)
    case(4096, 640, 640, res=True)
    case(16384, 1280, 320)                                  # ff.net.2 backward-data shape (without its GEGLU-backward epilogue)
    case(1000, 320, 320, res=True, nb=1, force="64")        # ragged M
    case(300, 1280, 640, geglu=True, nb=3, force="64")      # ragged M, GEGLU
    print("ALL PASS" if ok_all else "SOME FAILED")
