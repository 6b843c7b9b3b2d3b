"""Row-resident chain kernel (aql_lora_chain_fwd, csrc/aql_chain.hip) against the launch sequence it replaces --
aql_lora_gemm_fused (+ residual) -> aql_layernorm_fwd -> aql_lora_gemm_fused x n -- on the twin-batch shapes of the 64 x 64 level:
every output must be BIT-IDENTICAL.  `time` as argv[1]: HIP-graph timings of both forms (20 launches per graph, min of 5 replays)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from aqualora_amd import _lib as L  # noqa: E402
from aqualora_amd import ops  # noqa: E402

dev = "cuda"
torch.manual_seed(0)
rnd = lambda *s, std=1.0: (torch.randn(*s, device=dev) * std).to(torch.bfloat16)  # noqa: E731
C = 320
TIME = len(sys.argv) > 1 and sys.argv[1] == "time"
ok_all = True


def lin(bias=True):
    return dict(W=rnd(C, C, std=C ** -0.5), bias=rnd(C, std=0.1) if bias else None, Ad=rnd(32, C, std=C ** -0.5), Bup=rnd(C, 32, std=0.2))


def unfused_linear(x, p, S, rps, row0, res, M):
    y = torch.empty(M, C, dtype=torch.bfloat16, device=dev)
    T = torch.full((M, 32), float("nan"), dtype=torch.bfloat16, device=dev)
    Ts = torch.full((M, 32), float("nan"), dtype=torch.bfloat16, device=dev)
    rc = L.call_raw("aql_lora_gemm_fused", L.ptr(x), C, L.ptr(p["W"]), C, M, C, C, L.ptr(p["Ad"]), L.ptr(S), rps, L.ptr(p["Bup"]),
                    L.ptr(p["bias"]), L.ptr(res), 0 if res is None else C, L.ptr(y), C, L.ptr(T), L.ptr(Ts), row0, L.stream_ptr())
    L.check(rc, "aql_lora_gemm_fused")
    return y, T, Ts


def unfused_ln(x, gamma, beta, M):
    y = torch.empty_like(x)
    st = torch.empty(M, 2, dtype=torch.float32, device=dev)
    L.call("aql_layernorm_fwd", L.ptr(x), M, C, L.ptr(gamma), L.ptr(beta), 1e-5, L.ptr(y), L.ptr(st), L.stream_ptr())
    return y, st


def graph_time(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1000 / reps)
    return best


def eq(a, b, row0=0):
    return torch.equal(a[row0:].view(torch.int16), b[row0:].view(torch.int16))


for name, M, nb, twin, nq in [("a: to_out+res -> LN -> to_q", 32768, 8, True, 1), ("d: proj_in -> LN -> q|k|v", 32768, 8, True, 3),
                              ("a, no twin", 16384, 4, False, 1), ("b': to_out+res -> LN", 32768, 8, True, 0),
                              ("a, 2 rounds", 65536, 16, True, 1)]:
    rps = M // nb
    row0 = M // 2 if twin else 0
    X, R = rnd(M, C), rnd(M, C)
    S = rnd(nb, 32)
    if twin:
        S[: nb // 2] = 0
    p0 = lin()
    qs = [lin(bias=False) for _ in range(nq)]
    gamma, beta = rnd(C, std=0.3) + 1, rnd(C, std=0.1)

    def run_unfused():
        hs, T0, Ts0 = unfused_linear(X, p0, S, rps, row0, R, M)
        n, st = unfused_ln(hs, gamma, beta, M)
        outs = [unfused_linear(n, q, S, rps, row0, None, M) for q in qs]
        return hs, T0, Ts0, n, st, outs

    hs2 = torch.empty(M, C, dtype=torch.bfloat16, device=dev)
    n2 = torch.full((M, C), float("nan"), dtype=torch.bfloat16, device=dev)
    st2 = torch.empty(M, 2, dtype=torch.float32, device=dev)
    T2 = [torch.full((M, 32), float("nan"), dtype=torch.bfloat16, device=dev) for _ in range(nq + 1)]
    Ts2 = [torch.full((M, 32), float("nan"), dtype=torch.bfloat16, device=dev) for _ in range(nq + 1)]
    q2 = [torch.empty(M, C, dtype=torch.bfloat16, device=dev) for _ in range(nq)]
    stages = [dict(p0, ldw=C, T=T2[0], Ts=Ts2[0], res=R, ldr=C, out=hs2, ldo=C, keep=1, ln=1, gamma=gamma, beta=beta, eps=1e-5,
                   stats=st2, nout=n2, ldn=C, nout_row0=(row0 if nq else 0))]
    for i, q in enumerate(qs):
        stages.append(dict(q, ldw=C, T=T2[i + 1], Ts=Ts2[i + 1], out=q2[i], ldo=C, keep=0))

    def run_chain():
        ops.chain_fwd(X, C, M, rps, row0, S, stages)

    hs, T0, Ts0, n, st, outs = run_unfused()
    run_chain()
    torch.cuda.synchronize()
    checks = {"hs": eq(hs, hs2), "T0": eq(T0, T2[0], row0), "Ts0": eq(Ts0, Ts2[0], row0), "ln": eq(n, n2, row0 if nq else 0),
              "stats": torch.equal(st, st2)}
    for i, (y, T, Ts) in enumerate(outs):
        checks[f"q{i}"] = eq(y, q2[i])
        checks[f"T{i + 1}"] = eq(T, T2[i + 1], row0)
        checks[f"Ts{i + 1}"] = eq(Ts, Ts2[i + 1], row0)
    ok = all(checks.values())
    ok_all &= ok
    bad = [k for k, v in checks.items() if not v]
    extra = ""
    if not ok:
        for k, (u, c) in {"hs": (hs, hs2), "ln": (n, n2)}.items():
            d = (u.float() - c.float()).abs()
            extra += f" | {k}: {int((d > 0).sum())} differ, max {float(d.max()):.3e}"
        for i, (y, _, _) in enumerate(outs):
            d = (y.float() - q2[i].float()).abs()
            extra += f" | q{i}: {int((d > 0).sum())} differ, max {float(d.max()):.3e}"
    line = f"{'PASS' if ok else 'FAIL'} chain [{name}] M{M}: bit-identical {ok} {bad}{extra}"
    if TIME:
        tu, tc = graph_time(run_unfused), graph_time(run_chain)
        line += f" | unfused {tu:.1f} us, chain {tc:.1f} us, ratio {tc / tu:.2f}"
    print(line, flush=True)
print("ALL PASS" if ok_all else "SOME FAILED")
