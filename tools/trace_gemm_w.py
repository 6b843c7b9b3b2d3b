"""Per-step timeline of the wave-specialised conv kernel (library built with -DAQL_TRACE_W): compute wave 0 and loader
wave 4 of workgroup 0."""
import sys, os
os.environ["AQL_TILE"] = os.environ.get("AQL_TILE", "11")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aqualora_amd import _lib as L
dev = "cuda"
ws = torch.zeros(16 << 20, dtype=torch.float32, device=dev)
rnd = lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16)
Bn, H, Cin, Cout = 4, 64, int(sys.argv[1]) if len(sys.argv) > 1 else 320, 320
if len(sys.argv) > 4: Bn, H, Cout = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])   # e.g. 1280 8 8 1280 with AQL_SPLITS=1
xh = rnd(Bn, H, H, Cin); wk = rnd(Cout, 9 * Cin); b = rnd(Cout); y = torch.empty(Bn, H, H, Cout, dtype=torch.bfloat16, device=dev)
call = lambda: L.call("aql_conv3x3_fwd", L.ptr(xh), Bn, H, H, Cin, L.ptr(wk), L.ptr(b), Cout, 1, 0, None, 0, None, L.ptr(y),
                      L.ptr(ws), ws.numel() * 4, L.stream_ptr())
for _ in range(3): call()
torch.cuda.synchronize()
t = ws.view(torch.int64)[:16 * 1024].cpu().view(16, 256, 4)
nst = min(9 * Cin // 64, 250)
for wv in (0, 3, 4, 7, 8, 12):
    tr = t[wv, :nst].double()
    top, a1, a2, a3 = tr[:, 0], tr[:, 1], tr[:, 2], tr[:, 3]
    nxt = torch.cat([top[1:], top[-1:]])
    role = "loader " if (wv % 8) >= 4 else "compute"
    if role == "loader ":
        print(f"wave {wv} {role}: wait-landed {float((a1-top)[5:-1].mean()):.0f}  barrier {float((a2-a1)[5:-1].mean()):.0f}  issue {float((a3-a2)[5:-1].mean()):.0f}  step {float((nxt-top)[5:-1].mean()):.0f}")
    else:
        print(f"wave {wv} {role}: barrier {float((a2-top)[5:-1].mean()):.0f}  compute {float((nxt-a2)[5:-1].mean()):.0f}  step {float((nxt-top)[5:-1].mean()):.0f}")
