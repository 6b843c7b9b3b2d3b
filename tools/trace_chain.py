"""Phase timeline of one workgroup of the row-resident chain kernel (aql_lora_chain_fwd): cycle stamps of wavefronts 0 and 4 of
every block, written when AQL_CHAIN_TRACE_BUF names a device buffer.  Prints the median over blocks of each phase's length.
Needs a trace build of the library (the product build has no stamps):
    tools/build_alt.sh trace aql_chain.hip -DAQL_CHAIN_TRACE=1  &&  AQL_LIB=altlib/trace.so python tools/trace_chain.py 1 [wide]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from aqualora_amd import ops  # noqa: E402

dev = "cuda"
torch.manual_seed(0)
rnd = lambda *s, std=1.0: (torch.randn(*s, device=dev) * std).to(torch.bfloat16)  # noqa: E731
C, M, nb = 320, 32768, 8
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 1
WIDE = len(sys.argv) > 2 and sys.argv[2] == "wide"      # the rank-320 kernel (aql_lora_chain_fwd_r320), config 3's twin batch
if WIDE:
    M, nb = 65536, 16
RK = 320 if WIDE else 32
rps, row0 = M // nb, M // 2
X, R, S = rnd(M, C), rnd(M, C), rnd(nb, RK)
S[: nb // 2] = 0
lin = lambda bias: dict(W=rnd(C, C, std=C ** -0.5), bias=rnd(C, std=0.1) if bias else None, Ad=rnd(RK, C, std=C ** -0.5), Bup=rnd(C, RK, std=0.2 if RK == 32 else 0.05), ldw=C)  # noqa: E731
mk = lambda *s: torch.empty(*s, dtype=torch.bfloat16, device=dev)  # noqa: E731
gamma, beta = rnd(C, std=0.3) + 1, rnd(C, std=0.1)
stages = [dict(lin(True), T=mk(M, RK), Ts=mk(M, RK), res=R, ldr=C, out=mk(M, C), ldo=C, keep=1, ln=1, gamma=gamma, beta=beta, eps=1e-5,
               stats=torch.empty(M, 2, device=dev), nout=mk(M, C), ldn=C, nout_row0=row0 if nq else 0)]
for _ in range(nq):
    stages.append(dict(lin(False), T=mk(M, RK), Ts=mk(M, RK), out=mk(M, C), ldo=C, keep=0))
nblk = M // (64 if WIDE else 128)
buf = torch.zeros(nblk * 2 * 32, dtype=torch.int64, device=dev)
for _ in range(3):
    ops.chain_fwd(X, C, M, rps, row0, S, stages, rank=RK)
torch.cuda.synchronize()
os.environ["AQL_CHAIN_TRACE_BUF"] = str(buf.data_ptr())
ops.chain_fwd(X, C, M, rps, row0, S, stages, rank=RK)
torch.cuda.synchronize()
del os.environ["AQL_CHAIN_TRACE_BUF"]
t = buf.view(nblk, 2, 32).cpu()
names = ["start"]
for g in range(nq + 1):
    if WIDE:
        names += [f"g{g} pass T (10 tiles)", f"g{g} T / Ts out", f"g{g} X.W (10 tiles)", f"g{g} Ts.Bup (10 tiles)", f"g{g} epilogue", f"g{g} row pass"]
    else:
        names += [f"g{g} tile0 landed", f"g{g} K loop", f"g{g} up step", f"g{g} epilogue", f"g{g} published", f"g{g} row pass"]
names.append("drained")
kinds = (("LoRA tiles", slice(0, nblk // 2)), ("clean tiles", slice(nblk // 2, nblk))) if WIDE else \
    (("clean tiles", slice(0, nblk, 2)), ("LoRA tiles", slice(1, nblk, 2)))
for kind, sel in kinds:
    for w in (0, 1):
        tt = t[sel, w]
        d = (tt[:, 1:len(names)] - tt[:, :len(names) - 1]).double()
        tot = (tt[:, len(names) - 1] - tt[:, 0]).double()
        print(f"-- {kind}, wavefront {w * 4}: total {tot.median():.0f} cycles (min {tot.min():.0f}, max {tot.max():.0f})")
        for i in range(1, len(names)):
            print(f"   {names[i]:<24} {d[:, i - 1].median():8.0f}   (min {d[:, i - 1].min():.0f}, max {d[:, i - 1].max():.0f})")
span = (t[:, :, len(names) - 1].max() - t[:, :, 0].min())
print(f"first start -> last drained: {int(span)} cycles (s_memtime ticks at 100 MHz? see DESIGN)")
