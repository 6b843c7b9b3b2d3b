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


def lin(bias=True, lora=True):
    return dict(W=rnd(C, C, std=C ** -0.5), bias=rnd(C, std=0.1) if bias else None, Ad=rnd(32, C, std=C ** -0.5) if lora else None,
                Bup=rnd(C, 32, std=0.2) if lora else None)


def unfused_linear(x, p, S, rps, row0, res, M):
    y = torch.empty(M, C, dtype=torch.bfloat16, device=dev)
    T = torch.full((M, 32), float("nan"), dtype=torch.bfloat16, device=dev)
    Ts = torch.full((M, 32), float("nan"), dtype=torch.bfloat16, device=dev)
    if p["Ad"] is None:     # the LoRA-free linear of the clean / inference passes
        ops.gemm_bf16(x, p["W"], p["bias"], residual=res, out=y)
        return y, T, Ts
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
                              ("a, 2 rounds", 65536, 16, True, 1), ("a, batch-1 twin (64-row tiles)", 8192, 2, True, 1),
                              ("d, no LoRA, CFG batch 2 (64-row tiles)", 8192, 2, False, 3), ("a, no LoRA, CFG batch 2", 8192, 2, False, 1)]:
    rps = M // nb
    row0 = M // 2 if twin else 0
    nolora = "no LoRA" in name
    X, R = rnd(M, C), rnd(M, C)
    S = rnd(nb, 32)
    if twin:
        S[: nb // 2] = 0
    p0 = lin(lora=not nolora)
    qs = [lin(bias=False, lora=not nolora) for _ in range(nq)]
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
    checks = {"hs": eq(hs, hs2), "ln": eq(n, n2, row0 if nq else 0), "stats": torch.equal(st, st2)}
    if not nolora:
        checks.update(T0=eq(T0, T2[0], row0), Ts0=eq(Ts0, Ts2[0], row0))
    for i, (y, T, Ts) in enumerate(outs):
        checks[f"q{i}"] = eq(y, q2[i])
        if not nolora:
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
print("forward chains: " + ("ok" if ok_all else "SOME FAILED"))

# ---------------------------------------------------------------------------------------------------------------- backward chains
# aql_lora_chain_bwd against  aql_layernorm_bwd / aql_lora_gemm_fused (backward-data operands)  in sequence, M = 16384 (64-row tiles)
print("backward chains:", flush=True)
ok_b = True
for name, M, nb, pre, nst in [("B1: LN bwd -> to_out bwd", 16384, 4, True, 1), ("B2: to_q bwd -> LN bwd -> to_out bwd", 16384, 4, False, 2),
                              ("B1 without a residual gradient", 16384, 4, True, 1), ("B2, 8192 rows", 8192, 2, False, 2)]:
    rps = M // nb
    dY, Xs, dres, S = rnd(M, C), rnd(M, C), rnd(M, C), rnd(nb, 32)
    if "without" in name:
        dres = None
    gamma = rnd(C, std=0.3) + 1
    stats = torch.stack([torch.randn(M, device=dev) * 0.1, torch.rand(M, device=dev) + 0.5], dim=1).contiguous()
    lins = [dict(Wt=rnd(C, C, std=C ** -0.5), BupT=rnd(32, C, std=0.2), AT=rnd(C, 32, std=C ** -0.5)) for _ in range(nst)]

    def lin_bwd(x, p):   # dTs = dY.Bup, dT = dTs * S, dX = dY.W + dT.A : the one-launch kernel with exchanged operands
        y = torch.empty(M, C, dtype=torch.bfloat16, device=dev)
        T = torch.empty(M, 32, dtype=torch.bfloat16, device=dev)
        Ts = torch.empty_like(T)
        L.check(L.call_raw("aql_lora_gemm_fused", L.ptr(x), C, L.ptr(p["Wt"]), C, M, C, C, L.ptr(p["BupT"]), L.ptr(S), rps, L.ptr(p["AT"]),
                           None, None, 0, L.ptr(y), C, L.ptr(T), L.ptr(Ts), 0, L.stream_ptr()), "aql_lora_gemm_fused")
        return y, T, Ts

    def ln_bwd(dy):
        dx = torch.empty_like(dy)
        L.call("aql_layernorm_bwd", L.ptr(Xs), L.ptr(dy), M, C, L.ptr(gamma), L.ptr(stats), L.ptr(dres), L.ptr(dx), L.stream_ptr())
        return dx

    def run_unfused():
        if pre:
            dh = ln_bwd(dY)
            return [dh] + list(lin_bwd(dh, lins[0]))
        dn, T0, Ts0 = lin_bwd(dY, lins[0])
        dh = ln_bwd(dn)
        return [dh, T0, Ts0] + list(lin_bwd(dh, lins[1]))

    mk = lambda *s: torch.full(s, float("nan"), dtype=torch.bfloat16, device=dev)  # noqa: E731
    dh2 = mk(M, C)
    outs2 = [(mk(M, C), mk(M, 32), mk(M, 32)) for _ in range(nst)]
    ln_entry = dict(x=Xs, ldx=C, stats=stats, gamma=gamma, dres=dres, lddres=C, out=dh2, ldo=C)
    if pre:
        stages = [dict(lins[0], ldw=C, dTs=outs2[0][1], dT=outs2[0][2], dX=outs2[0][0], lddx=C, keep=0)]
        lns = [ln_entry, None]
    else:
        stages = [dict(lins[0], ldw=C, dTs=outs2[0][1], dT=outs2[0][2], keep=1),
                  dict(lins[1], ldw=C, dTs=outs2[1][1], dT=outs2[1][2], dX=outs2[1][0], lddx=C, keep=0)]
        lns = [None, ln_entry, None]

    def run_chain():
        ops.chain_bwd(dY, C, M, rps, S, stages, lns)

    ref = run_unfused()
    run_chain()
    torch.cuda.synchronize()
    if pre:
        got = [dh2, outs2[0][0], outs2[0][1], outs2[0][2]]
    else:
        got = [dh2, outs2[0][1], outs2[0][2], outs2[1][0], outs2[1][1], outs2[1][2]]
    same = [eq(a_, b_) for a_, b_ in zip(ref, got)]
    ok = all(same)
    ok_b &= ok
    line = f"{'PASS' if ok else 'FAIL'} chain [{name}] M{M}: bit-identical {same}"
    if not ok:
        for a_, b_ in zip(ref, got):
            d = (a_.float() - b_.float()).abs()
            line += f" | {int((d > 0).sum())} differ / nan {int(torch.isnan(b_.float()).sum())}"
    if TIME:
        tu, tc = graph_time(run_unfused), graph_time(run_chain)
        line += f" | unfused {tu:.1f} us, chain {tc:.1f} us, ratio {tc / tu:.2f}"
    print(line, flush=True)

# ---------------------------------------------------------------------------------------------------------------- rank-320 chains
# aql_lora_chain_fwd_r320 against  aql_lora_down (LoRA rows only) + aql_gemm_bf16 with the second K segment [Ts | Bup]  (+ residual) ->
# aql_layernorm_fwd -> ...  on the twin batch of BASELINE config 3 (batch 8: 65536 rows) and smaller ones
print("rank-320 chains:", flush=True)
ok_w = True
R3 = 320


def lin3(bias=True):
    return dict(W=rnd(C, C, std=C ** -0.5), bias=rnd(C, std=0.1) if bias else None, Ad=rnd(R3, C, std=C ** -0.5), Bup=rnd(C, R3, std=0.05))


def unfused3(x, p, S, rps, row0, res, M):
    y = torch.empty(M, C, dtype=torch.bfloat16, device=dev)
    T = torch.full((M, R3), float("nan"), dtype=torch.bfloat16, device=dev)
    Ts = torch.full((M, R3), float("nan"), dtype=torch.bfloat16, device=dev)
    ops.lora_down(x[row0:], C, M - row0, C, p["Ad"], R3, S[row0 // rps:], rps, T[row0:], Ts[row0:])
    ops.gemm_bf16(x, p["W"], p["bias"], A2=Ts, B2=p["Bup"], residual=res, out=y, lora_row0=row0)
    return y, T, Ts


for name, M, nb, twin, nq in [("a: to_out+res -> LN -> to_q", 65536, 16, True, 1), ("d: proj_in -> LN -> q|k|v", 65536, 16, True, 3),
                              ("b': to_out+res -> LN", 65536, 16, True, 0), ("a, no twin", 32768, 8, False, 1),
                              ("d, batch 4 twin", 32768, 8, True, 3)]:
    rps = M // nb
    row0 = M // 2 if twin else 0
    X, R = rnd(M, C), rnd(M, C)
    S = rnd(nb, R3)
    if twin:
        S[: nb // 2] = 0
    p0 = lin3()
    qs = [lin3(bias=False) for _ in range(nq)]
    gamma, beta = rnd(C, std=0.3) + 1, rnd(C, std=0.1)

    def run_unfused():
        hs, T0, Ts0 = unfused3(X, p0, S, rps, row0, R, M)
        n, st = unfused_ln(hs, gamma, beta, M)
        outs = [unfused3(n, q, S, rps, row0, None, M) for q in qs]
        return hs, T0, Ts0, n, st, outs

    hs2 = torch.empty(M, C, dtype=torch.bfloat16, device=dev)
    n2 = torch.full((M, C), float("nan"), dtype=torch.bfloat16, device=dev)
    st2 = torch.empty(M, 2, dtype=torch.float32, device=dev)
    T2 = [torch.full((M, R3), float("nan"), dtype=torch.bfloat16, device=dev) for _ in range(nq + 1)]
    Ts2 = [torch.full((M, R3), float("nan"), dtype=torch.bfloat16, device=dev) for _ in range(nq + 1)]
    q2 = [torch.empty(M, C, dtype=torch.bfloat16, device=dev) for _ in range(nq)]
    stages = [dict(p0, ldw=C, T=T2[0], Ts=Ts2[0], res=R, ldr=C, out=hs2, ldo=C, keep=1, ln=1, gamma=gamma, beta=beta, eps=1e-5,
                   stats=st2, nout=n2, ldn=C, nout_row0=(row0 if nq else 0))]
    for i, q in enumerate(qs):
        stages.append(dict(q, ldw=C, T=T2[i + 1], Ts=Ts2[i + 1], out=q2[i], ldo=C, keep=0))

    def run_chain():
        ops.chain_fwd(X, C, M, rps, row0, S, stages, rank=R3)

    hs, T0, Ts0, n, st, outs = run_unfused()
    run_chain()
    torch.cuda.synchronize()
    checks = {"hs": eq(hs, hs2), "ln": eq(n, n2, row0 if nq else 0), "stats": torch.equal(st, st2), "T0": eq(T0, T2[0], row0),
              "Ts0": eq(Ts0, Ts2[0], row0)}
    for i, (y, T, Ts) in enumerate(outs):
        checks[f"q{i}"] = eq(y, q2[i])
        checks[f"T{i + 1}"] = eq(T, T2[i + 1], row0)
        checks[f"Ts{i + 1}"] = eq(Ts, Ts2[i + 1], row0)
    ok = all(checks.values())
    ok_w &= ok
    bad = [k for k, v in checks.items() if not v]
    extra = ""
    if not ok:
        for k, (u, c) in {"hs": (hs, hs2), "ln": (n, n2), "T0": (T0[row0:], T2[0][row0:]), "Ts0": (Ts0[row0:], Ts2[0][row0:])}.items():
            d = (u.float() - c.float()).abs()
            extra += f" | {k}: {int((d > 0).sum())} differ, max {float(d.nan_to_num(9e9).max()):.3e}"
        for i, (y, _, _) in enumerate(outs):
            d = (y.float() - q2[i].float()).abs()
            extra += f" | q{i}: {int((d > 0).sum())} differ, max {float(d.nan_to_num(9e9).max()):.3e}"
    line = f"{'PASS' if ok else 'FAIL'} chain r320 [{name}] M{M}: bit-identical {ok} {bad}{extra}"
    if TIME:
        tu, tc = graph_time(run_unfused), graph_time(run_chain)
        line += f" | unfused {tu:.1f} us, chain {tc:.1f} us, ratio {tc / tu:.2f}"
    print(line, flush=True)

print("ALL PASS" if (ok_all and ok_b and ok_w) else "SOME FAILED")
