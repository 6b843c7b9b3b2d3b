import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from aqualora_amd import ops
torch.manual_seed(0)
def run(B, H, d, Nq, Nk):
    q = torch.randn(B, Nq, H * d, device="cuda").to(torch.bfloat16).requires_grad_(True)
    k = torch.randn(B, Nk, H * d, device="cuda").to(torch.bfloat16).requires_grad_(True)
    v = torch.randn(B, Nk, H * d, device="cuda").to(torch.bfloat16).requires_grad_(True)
    do = torch.randn(B, Nq, H * d, device="cuda").to(torch.bfloat16)
    o = ops.attention(q, k, v, H)
    o.backward(do)
    qf, kf, vf = (t.detach().float().view(B, -1, H, d).transpose(1, 2).requires_grad_(True) for t in (q, k, v))
    of = torch.nn.functional.scaled_dot_product_attention(qf, kf, vf)
    of.backward(do.float().view(B, Nq, H, d).transpose(1, 2))
    def err(a, b):
        b = b.transpose(1, 2).reshape(a.shape)
        return float((a.float() - b).abs().max() / b.abs().max()) if torch.isfinite(a).all() else float("nan")
    print(f"B={B} H={H} d={d} Nq={Nq} Nk={Nk}: o {err(o, of):.3e} dq {err(q.grad, qf.grad):.3e} dk {err(k.grad, kf.grad):.3e} dv {err(v.grad, vf.grad):.3e}", flush=True)
for args in [(2, 2, 80, 256, 77), (2, 8, 80, 1024, 77), (2, 2, 80, 1024, 77), (2, 2, 80, 256, 256), (2, 2, 160, 64, 77), (2, 2, 160, 16, 77), (2, 2, 160, 4, 77),
             (2, 2, 160, 4, 4), (2, 2, 160, 16, 16), (2, 8, 40, 256, 77), (1, 2, 80, 256, 77), (2, 2, 80, 128, 77), (2, 2, 80, 512, 77)]:
    run(*args)
