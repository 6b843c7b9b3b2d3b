"""Determinism of the chain kernel: the same launch N times, every output compared bit for bit with the first run."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from aqualora_amd import ops  # noqa: E402

dev = "cuda"
torch.manual_seed(1)
rnd = lambda *s, std=1.0: (torch.randn(*s, device=dev) * std).to(torch.bfloat16)  # noqa: E731
C, M, nb = 320, 32768, 8
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rps, row0 = M // nb, M // 2
lin = lambda bias: dict(W=rnd(C, C, std=C ** -0.5), bias=rnd(C, std=0.1) if bias else None, Ad=rnd(32, C, std=C ** -0.5), Bup=rnd(C, 32, std=0.2), ldw=C)  # noqa: E731
mk = lambda *s: torch.zeros(*s, dtype=torch.bfloat16, device=dev)  # noqa: E731
for nq, use_res in ((1, True), (3, False), (0, True)):
    X, R, S = rnd(M, C), rnd(M, C), rnd(nb, 32)
    S[: nb // 2] = 0
    gamma, beta = rnd(C, std=0.3) + 1, rnd(C, std=0.1)
    outs = dict(T0=mk(M, 32), Ts0=mk(M, 32), hs=mk(M, C), n=mk(M, C), st=torch.zeros(M, 2, device=dev))
    stages = [dict(lin(True), T=outs["T0"], Ts=outs["Ts0"], res=R if use_res else None, ldr=C, out=outs["hs"], ldo=C, keep=1, ln=1, gamma=gamma,
                   beta=beta, eps=1e-5, stats=outs["st"], nout=outs["n"], ldn=C, nout_row0=row0 if nq else 0)]
    for i in range(nq):
        outs[f"q{i}"], outs[f"T{i + 1}"], outs[f"Ts{i + 1}"] = mk(M, C), mk(M, 32), mk(M, 32)
        stages.append(dict(lin(False), T=outs[f"T{i + 1}"], Ts=outs[f"Ts{i + 1}"], out=outs[f"q{i}"], ldo=C, keep=0))
    ops.chain_fwd(X, C, M, rps, row0, S, stages)
    torch.cuda.synchronize()
    ref = {k: v.clone() for k, v in outs.items()}
    bad = {}
    for it in range(N):
        for v in outs.values():
            v.zero_()
        ops.chain_fwd(X, C, M, rps, row0, S, stages)
        torch.cuda.synchronize()
        for k, v in outs.items():
            if not torch.equal(v, ref[k]):
                d = (v.float() - ref[k].float()).abs()
                rows = torch.nonzero(d.reshape(M, -1).amax(1) > 0).flatten()
                bad.setdefault(k, []).append((it, int((d > 0).sum()), rows[:6].tolist()))
    print(f"nq={nq} res={use_res}: {N} repeats, mismatching outputs: " + (str({k: (len(v), v[:3]) for k, v in bad.items()}) if bad else "none"), flush=True)
