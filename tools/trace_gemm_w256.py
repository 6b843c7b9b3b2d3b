"""Per-K-tile timeline of the 12-wave 256x160 conv kernel (library built with -DAQL_TRACE_W, selected with AQL_LIB; AQL_TILE=14):
compute wavefronts 0 and 7, loader wavefronts 8 and 11 of workgroup 0.  Cycles of clock64 per K tile."""
import sys, os
os.environ["AQL_TILE"] = "14"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aqualora_amd import _lib as L
dev = "cuda"
ws = torch.zeros(16 << 20, dtype=torch.float32, device=dev)
rnd = lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16)
Bn, H, Cin, Cout = 8, 64, int(sys.argv[1]) if len(sys.argv) > 1 else 320, 320
xh = rnd(Bn, H, H, Cin); wk = rnd(Cout, 9 * Cin); b = rnd(Cout); y = torch.empty(Bn, H, H, Cout, dtype=torch.bfloat16, device=dev)
call = lambda: L.call("aql_conv3x3_fwd", L.ptr(xh), Bn, H, H, Cin, L.ptr(wk), L.ptr(b), Cout, 1, 0, None, 0, None, L.ptr(y),
                      L.ptr(ws), ws.numel() * 4, L.stream_ptr())
for _ in range(3): call()
torch.cuda.synchronize()
t = ws.view(torch.int64)[:24 * 1024].cpu().view(24, 256, 4)
nst = 9 * Cin // 64
for wv in (0, 7, 8, 11):
    tr = t[wv, :nst].double()
    top, a1, a2, a3 = tr[:, 0], tr[:, 1], tr[:, 2], tr[:, 3]
    nxt = torch.cat([top[1:], top[-1:]])
    if wv >= 8:
        print(f"wave {wv} loader : wait-landed {float((a1-top)[5:-1].mean()):.0f}  barrier {float((a2-a1)[5:-1].mean()):.0f}  issue {float((a3-a2)[5:-1].mean()):.0f}  step {float((nxt-top)[5:-1].mean()):.0f}")
    else:
        print(f"wave {wv} compute: barrier {float((a2-top)[5:-1].mean()):.0f}  compute {float((nxt-a2)[5:-1].mean()):.0f}  step {float((nxt-top)[5:-1].mean()):.0f}")
print(f"whole K loop of wave 0: {float(t[0, nst - 1, 0] - t[0, 0, 0]):.0f} cycles for {nst - 1} tiles")
