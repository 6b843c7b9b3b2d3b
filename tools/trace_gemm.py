"""Per-step timeline of one conv workgroup (AQL_ABL=9 build probe): where a K step's time goes inside the kernel."""
import sys, os
os.environ["AQL_ABL"] = "9"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aqualora_amd import _lib as L
dev = "cuda"
ws = torch.zeros(16 << 20, dtype=torch.float32, device=dev)
rnd = lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16)
Bn, H, Cin, Cout = 4, 64, int(sys.argv[1]) if len(sys.argv) > 1 else 320, 320
xh = rnd(Bn, H, H, Cin); wk = rnd(Cout, 9 * Cin); b = rnd(Cout); y = torch.empty(Bn, H, H, Cout, dtype=torch.bfloat16, device=dev)
call = lambda: L.call("aql_conv3x3_fwd", L.ptr(xh), Bn, H, H, Cin, L.ptr(wk), L.ptr(b), Cout, 1, 0, None, 0, None, L.ptr(y),
                      L.ptr(ws), ws.numel() * 4, L.stream_ptr())
for _ in range(3): call()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); call(); e1.record(); torch.cuda.synchronize()
print(f"kernel+launch {e0.elapsed_time(e1)*1e3:.1f} us")
t = ws.view(torch.int64)[:8 * 1024].cpu().view(8, 256, 4)
nst = 9 * Cin // 64
for wv in (0, 3, 4):
    tr = t[wv, :nst].double()
    t0 = tr[0, 0]
    top, landed, passed, issued = tr[:, 0], tr[:, 1], tr[:, 2], tr[:, 3]
    nxt = torch.cat([top[1:], top[-1:]])
    print(f"wave slot {wv}: ticks per step (mean over steps 5..{nst-2}): wait-data {float((landed-top)[5:-1].mean()):.0f}  barrier {float((passed-landed)[5:-1].mean()):.0f}  "
          f"dma-issue {float((issued-passed)[5:-1].mean()):.0f}  compute {float((nxt-issued)[5:-1].mean()):.0f}  total {float((nxt-top)[5:-1].mean()):.0f}   whole loop {float(top[-1]-t0):.0f} ticks")
print("first 8 steps of wave 0 (relative ticks):")
tr = t[0, :8].double(); print((tr - tr[0, 0]).long().tolist())
