"""Per-column-tile timeline of the strip kernel (library built with -DAQL_TRACE_L, selected with AQL_LIB): compute wave 0 of
every workgroup stamps the start of each column tile, the end of its K loop, the end of the Bup step and the end of its epilogue."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aqualora_amd import ops
dev = "cuda"
rnd = lambda *s, std=1.0: (torch.randn(*s, device=dev) * std).to(torch.bfloat16)
class Site:
    def __init__(self, r, K, N):
        self.rank = r
        self.a16, self.b16 = rnd(r, K, std=K ** -0.5), rnd(N, r, std=0.1)
        self.at16, self.bt16 = self.a16.t().contiguous(), self.b16.t().contiguous()
class Grab:
    def save_for_backward(self, *t): self.saved = t
def run(M, N, K, geglu, res=False, nb=4):
    pk = ops.PackedLinear(torch.randn(N, K, device=dev) * K ** -0.5, torch.randn(N, device=dev) * 0.1)
    site = Site(32, K, N)
    x = rnd(M, K); S16 = (1.0 + 0.3 * torch.randn(nb, 32, device=dev)).to(torch.bfloat16)
    r = rnd(M, N) if res else None
    buf = torch.zeros(1 << 20, dtype=torch.int64, device=dev)
    os.environ["AQL_TRACE_BUF"] = hex(buf.data_ptr())
    for _ in range(3):
        with torch.no_grad(): ops.LoraLinearFn.forward(Grab(), x, pk, site, S16, S16, M // nb, r, geglu, True)
    torch.cuda.synchronize(); buf.zero_()
    with torch.no_grad(): ops.LoraLinearFn.forward(Grab(), x, pk, site, S16, S16, M // nb, r, geglu, True)
    torch.cuda.synchronize()
    t = buf.view(-1, 128).cpu(); t = t[t[:, 0] != 0].double()
    ntn = (N // 2 if geglu else N) // (80 if geglu else 160)
    n = min(ntn, 30)
    st = t[:, 4:4 + 4 * n].view(-1, n, 4)
    print(f"M={M} N={N} K={K} geglu={geglu} res={res}: {t.shape[0]} workgroups, life {float((t[:,2]-t[:,0]).mean()):.0f} cyc; X panel + T phase {float((t[:,1]-t[:,0]).mean()):.0f}; "
          f"per column tile: K loop {float((st[:,:,1]-st[:,:,0]).mean()):.0f}  Bup step {float((st[:,:,2]-st[:,:,1]).mean()):.0f}  epilogue {float((st[:,:,3]-st[:,:,2]).mean()):.0f}  "
          f"(tile to tile {float((st[:,1:,0]-st[:,:-1,0]).mean()):.0f})")
torch.manual_seed(0)
run(16384, 2560, 320, True)
run(32768, 2560, 320, True, nb=8)
run(32768, 320, 320, False, res=True, nb=8)
run(32768, 960, 320, False, nb=8)
run(8192, 640, 640, False, res=True, nb=8)
