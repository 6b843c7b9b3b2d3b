"""Which tensor of the chained forward differs between two identical steps?  Records every ChainFn / attention output of two runs."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from aqualora_amd import ops, synth  # noqa: E402
from aqualora_amd.lora import inject_lora  # noqa: E402
from aqualora_amd.ppft import PPFTTrainer  # noqa: E402
from aqualora_amd.unet import UNet2DConditionModel, init_synthetic, lora_keys  # noqa: E402
from aqualora_amd.watermark import MapperNet, SecretEncoder  # noqa: E402

DEV = "cuda"
seed, rank, B = 4096, 32, 4
unet = UNet2DConditionModel(device=DEV, dtype=torch.bfloat16)
init_synthetic(unet, seed)
keys = lora_keys(unet)
inject_lora(unet, rank, keys)
with torch.no_grad():
    for k in keys:
        lay = unet.get_submodule(k).lora_layer
        lay.down.weight.copy_(synth.normal(k + ".lora.down", lay.down.weight.shape, 1.0 / rank, seed, DEV))
        lay.up.weight.copy_(synth.normal(k + ".lora.up", lay.up.weight.shape, 0.02, seed, DEV))
mapper = MapperNet(48, rank)
tr = PPFTTrainer(unet, mapper, SecretEncoder(48), rank)
z = synth.normal("b4.z", (B, 4, 64, 64), 1.0, seed).to(DEV)
wm = synth.normal("b4.wm", (B, 4, 64, 64), 0.05, seed).to(DEV)
eps = synth.normal("b4.eps", (B, 4, 64, 64), 1.0, seed).to(DEV)
msg = synth.bits("b4.msg", (B, 48), seed).to(DEV)
ctx = synth.normal("b4.ctx", (B, 77, 768), 1.0, seed).to(DEV).to(torch.bfloat16)
t = torch.tensor([500, 20, 981, 333], device=DEV)
tr.sec_encoder.encode = lambda m, out_scale=1.0: wm
rec = []
orig_chain, orig_attn = ops.lora_chain, ops.attention


def chain(x2d, res, S, S16, rps, stages):
    outs = orig_chain(x2d, res, S, S16, rps, stages)
    full_in = ops._full(x2d)
    rec.append(("chain_in", (full_in if full_in is not None else x2d).detach().clone()))
    for i, o in enumerate(outs):
        f = ops._full(o)
        rec.append((f"chain{len(stages)}_out{i}", (f if f is not None else o).detach().clone()))
    return outs


def attn(q, k, v, heads):
    o = orig_attn(q, k, v, heads)
    f = ops._full(o)
    rec.append(("attn", (f if f is not None else o).detach().clone()))
    return o


ops.lora_chain, ops.attention = chain, attn
runs = []
for r in range(3):
    rec.clear()
    tr.bank.zero_grad()
    loss, pred, clean = tr.forward_backward(z, msg, eps, t, ctx)
    torch.cuda.synchronize()
    runs.append([(n, x) for n, x in rec] + [("pred", pred.detach().clone())])
for r in (1, 2):
    for i, ((n, a), (_, b)) in enumerate(zip(runs[0], runs[r])):
        if not torch.equal(a, b):
            d = (a.float() - b.float()).abs()
            rows = torch.nonzero(d.reshape(-1, d.shape[-1]).amax(1) > 0).flatten()
            print(f"run {r}: first difference at record {i} ({n}), shape {tuple(a.shape)}: {int((d > 0).sum())} elements, rows {rows[:8].tolist()} ... {rows[-3:].tolist()}, n rows {len(rows)}")
            for rw in rows[:4].tolist():
                cols = torch.nonzero(d.reshape(-1, d.shape[-1])[rw] > 0).flatten().tolist()
                print(f"   row {rw} (tile {rw // 128}, row in tile {rw % 128}): cols {cols}; a {a.reshape(-1, a.shape[-1])[rw, cols[:4]].tolist()} b {b.reshape(-1, b.shape[-1])[rw, cols[:4]].tolist()}")
            break
    else:
        print(f"run {r}: identical ({len(runs[0])} records)")
