"""debug: where do non-finite gradients come from on the 160/320-channel tiny U-Net?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests.common import T, ppft_inputs, tiny_lora, tiny_unet
from aqualora_amd.lora import LoraBank, inject_lora, patch_lora_forwards
from aqualora_amd.unet import lora_keys, BasicTransformerBlock, ResnetBlock2D, Attention, FeedForward
DEV = "cuda"
rank = 32
cfg = dict(block_out_channels=(160, 320, 320, 320), cross_attention_dim=32, attention_heads=2, layers_per_block=1)
unet = tiny_unet(DEV, torch.bfloat16, cfg)
keys = lora_keys(unet)
state = {}
for k, (d, u) in tiny_lora(keys, unet, rank, 0.1).items():
    state[k + ".down.weight"], state[k + ".up.weight"] = d, u
inject_lora(unet, rank, keys, state)
patch_lora_forwards(unet)
if os.environ.get("BANK", "1") == "1":
    bank = LoraBank(unet)
inp = ppft_inputs(cfg, rank=rank, device=DEV)
x, t, ctx = inp["z"].to(torch.bfloat16), inp["t"], inp["ctx"].to(torch.bfloat16)
S = (1.0 + 0.3 * T("alpha.S", (x.shape[0], rank), 1.0, DEV)).requires_grad_(True)
names = {id(m): n for n, m in unet.named_modules()}
def hook(mod, gin, gout):
    fi = [bool(torch.isfinite(g).all()) for g in gin if g is not None]
    fo = [bool(torch.isfinite(g).all()) for g in gout if g is not None]
    if not all(fi) or not all(fo):
        print("NONFINITE", names[id(mod)], type(mod).__name__, "grad_out finite", fo, "grad_in finite", fi, flush=True)
for m in unet.modules():
    if isinstance(m, (BasicTransformerBlock, ResnetBlock2D, Attention, FeedForward)):
        m.register_full_backward_hook(hook)
y = unet(x, t, ctx, cross_attention_kwargs={"scale": S}).sample
print("y finite", bool(torch.isfinite(y).all()), float(y.float().abs().max()))
y.float().square().mean().backward()
print("S.grad finite", bool(torch.isfinite(S.grad).all()), S.grad.flatten()[:6])
for k in keys:
    lay = unet.get_submodule(k).lora_layer
    for nm, p in (("down", lay.down.weight), ("up", lay.up.weight)):
        if p.grad is not None and not torch.isfinite(p.grad).all():
            print("nonfinite weight grad", k, nm)
