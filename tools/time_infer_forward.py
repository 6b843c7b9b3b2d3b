import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from aqualora_amd import ops, synth
from aqualora_amd.unet import UNet2DConditionModel, init_synthetic
dev = "cuda"
unet = UNet2DConditionModel(device=dev, dtype=torch.bfloat16)
init_synthetic(unet, 2048)
x = synth.normal("z", (2, 4, 64, 64), 1.0, 1, dev).to(torch.bfloat16)
ctx = synth.normal("c", (2, 77, 768), 1.0, 1, dev).to(torch.bfloat16)
t = torch.tensor([500, 500], device=dev)
calls = [0]
orig = ops.chain_fwd
def counted(*a, **k):
    calls[0] += 1
    return orig(*a, **k)
ops.chain_fwd = counted
outs = {}
for chain in (False, True, False, True):
    ops.CHAIN = chain
    calls[0] = 0
    with torch.no_grad():
        for _ in range(3):
            y = unet(x, t, ctx, cross_attention_kwargs={"scale": None}).sample
        torch.cuda.synchronize()
        n0 = calls[0]
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            y = unet(x, t, ctx, cross_attention_kwargs={"scale": None}).sample
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            g.replay()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
    outs[chain] = y.clone()
    print(f"CHAIN={chain}: chain launches per forward {n0 // 3}, forward {dt * 1e3:.3f} ms")
print("bit-identical:", torch.equal(outs[False], outs[True]))
