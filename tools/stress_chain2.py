"""Determinism of the chain kernel under back-to-back launches with other kernels in between (no host sync).
argv[2] = "wide": the rank-320 kernel (aql_lora_chain_fwd_r320) on config 3's 65536-row twin batch."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from aqualora_amd import ops  # noqa: E402

dev = "cuda"
torch.manual_seed(1)
rnd = lambda *s, std=1.0: (torch.randn(*s, device=dev) * std).to(torch.bfloat16)  # noqa: E731
C, M, nb = 320, 32768, 8
N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
WIDE = len(sys.argv) > 2 and sys.argv[2] == "wide"
RK = 320 if WIDE else 32
if WIDE:
    M, nb = 65536, 16
rps, row0 = M // nb, M // 2
lin = lambda bias: dict(W=rnd(C, C, std=C ** -0.5), bias=rnd(C, std=0.1) if bias else None, Ad=rnd(RK, C, std=C ** -0.5), Bup=rnd(C, RK, std=0.2 if RK == 32 else 0.05), ldw=C)  # noqa: E731
mk = lambda *s: torch.zeros(*s, dtype=torch.bfloat16, device=dev)  # noqa: E731
big = torch.randn(64 << 20, device=dev)
for nq, use_res in ((3, False), (1, True), (0, True)):
    X, R, S = rnd(M, C), rnd(M, C), rnd(nb, RK)
    S[: nb // 2] = 0
    gamma, beta = rnd(C, std=0.3) + 1, rnd(C, std=0.1)
    lins = [lin(True)] + [lin(False) for _ in range(nq)]

    def make():
        outs = dict(T0=mk(M, RK), Ts0=mk(M, RK), hs=mk(M, C), n=mk(M, C), st=torch.zeros(M, 2, device=dev))
        stages = [dict(lins[0], T=outs["T0"], Ts=outs["Ts0"], res=R if use_res else None, ldr=C, out=outs["hs"], ldo=C, keep=1, ln=1,
                       gamma=gamma, beta=beta, eps=1e-5, stats=outs["st"], nout=outs["n"], ldn=C, nout_row0=row0 if nq else 0)]
        for i in range(nq):
            outs[f"q{i}"], outs[f"T{i + 1}"], outs[f"Ts{i + 1}"] = mk(M, C), mk(M, RK), mk(M, RK)
            stages.append(dict(lins[i + 1], T=outs[f"T{i + 1}"], Ts=outs[f"Ts{i + 1}"], out=outs[f"q{i}"], ldo=C, keep=0))
        return outs, stages

    sets = [make() for _ in range(4 if WIDE else 8)]
    ops.chain_fwd(X, C, M, rps, row0, S, sets[0][1], rank=RK)
    torch.cuda.synchronize()
    ref = {k: v.clone() for k, v in sets[0][0].items()}
    nbad = 0
    first = None
    for it in range(N):
        for outs, stages in sets:
            big.mul_(1.0001)                       # a bandwidth-bound kernel right in front
            ops.chain_fwd(X, C, M, rps, row0, S, stages, rank=RK)
        torch.cuda.synchronize()
        for outs, _ in sets:
            for k, v in outs.items():
                if not torch.equal(v, ref[k]):
                    nbad += 1
                    d = (v.float() - ref[k].float()).abs()
                    rows = torch.nonzero(d.reshape(M, -1).amax(1) > 0).flatten()
                    if first is None:
                        first = (it, k, int((d > 0).sum()), rows[:6].tolist())
                    if nbad <= 12:
                        rw = int(rows[0])
                        cols = torch.nonzero(d.reshape(M, -1)[rw] > 0).flatten().tolist()
                        print(f"  it {it} set {id(outs) % 1000} {k}: {int((d > 0).sum())} elements in rows {rows[:4].tolist()} cols {cols[:20]} got {v.reshape(M, -1)[rw, cols[:4]].tolist()} want {ref[k].reshape(M, -1)[rw, cols[:4]].tolist()}")
    print(f"nq={nq} res={use_res}: {N * len(sets)} launches, mismatching outputs: {nbad} first {first}", flush=True)
