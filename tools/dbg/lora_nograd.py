import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from aqualora_amd import lora as AL
torch.manual_seed(0)
DEV = "cuda"
def run(cin, cout, nb, n, r, xgrad):
    host = AL.LoRACompatibleLinear(cin, cout, bias=False, device=DEV, dtype=torch.bfloat16)
    ll = AL.LoRALinearLayer(cin, cout, r, device=DEV, dtype=torch.float32)
    with torch.no_grad():
        host.weight.copy_(torch.randn(cout, cin) * cin ** -0.5)
        ll.down.weight.copy_(torch.randn(r, cin) / r)
        ll.up.weight.copy_(torch.randn(cout, r) * 0.05)
    host.set_lora_layer(ll)
    x = torch.randn(nb, n, cin, device=DEV).to(torch.bfloat16).requires_grad_(xgrad)
    S = (torch.randn(nb, r, device=DEV) * 0.3 + 1).requires_grad_(True)
    y = AL.CustomLoRACompatibleLinearforward(host, x, S)
    dy = torch.randn_like(y)
    y.backward(dy)
    xf = x.detach().float()
    W, A, Bu = host.weight.float(), ll.down.weight.detach().clone().requires_grad_(True), ll.up.weight.detach().clone().requires_grad_(True)
    S2 = S.detach().clone().requires_grad_(True)
    yr = xf @ W.t() + ((xf @ A.t()) * S2[:, None, :]) @ Bu.t()
    yr.backward(dy.float())
    e = lambda a, b: float((a.float() - b).abs().max() / b.abs().max()) if torch.isfinite(a).all() else float("nan")
    print(f"cin={cin} cout={cout} nb={nb} n={n} r={r} xgrad={xgrad}: y {e(y, yr):.2e} dS {e(S.grad, S2.grad):.2e} ddown {e(ll.down.weight.grad, A.grad):.2e} dup {e(ll.up.weight.grad, Bu.grad):.2e}", flush=True)
for a in [(32, 160, 2, 77, 32, False), (32, 160, 2, 77, 32, True), (160, 160, 2, 256, 32, False), (160, 160, 2, 256, 32, True), (32, 64, 2, 77, 8, False),
          (768, 320, 4, 77, 32, False), (320, 320, 2, 256, 32, False), (32, 320, 2, 77, 32, False), (64, 160, 2, 77, 32, False), (32, 160, 2, 64, 32, False), (32, 160, 1, 77, 32, False)]:
    run(*a)
