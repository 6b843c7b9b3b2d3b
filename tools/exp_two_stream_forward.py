"""Experiment (round 5): the guided forward of config 4 (CFG batch 2, LoRA-free U-Net) as ONE batch-2 forward against TWO batch-1
forwards on two streams inside one HIP graph (parallel branches).  At batch 2 the forward is 445 launches of ~12 us on grids that
fill half the chip: does the second queue hide the launch floor?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aqualora_amd import synth
from aqualora_amd.unet import UNet2DConditionModel, init_synthetic
dev = "cuda"
unet = UNet2DConditionModel(device=dev, dtype=torch.bfloat16)
init_synthetic(unet, 2048)
x = synth.normal("z", (2, 4, 64, 64), 1.0, 1, dev).to(torch.bfloat16)
ctx = synth.normal("c", (2, 77, 768), 1.0, 1, dev).to(torch.bfloat16)
t = torch.tensor([500], device=dev)
xs = [x[i:i + 1].contiguous() for i in range(2)]
cs = [ctx[i:i + 1].contiguous() for i in range(2)]
kw = dict(cross_attention_kwargs={"scale": None})


def fwd2():
    return unet(x, t, ctx, **kw).sample


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def fwd_split():
    cur = torch.cuda.current_stream()
    ev = torch.cuda.Event()
    ev.record(cur)
    outs = [None, None]
    for i, s in enumerate((s1, s2)):
        s.wait_event(ev)
        with torch.cuda.stream(s):
            outs[i] = unet(xs[i], t, cs[i], **kw).sample
    for s in (s1, s2):
        e = torch.cuda.Event()
        e.record(s)
        cur.wait_event(e)
    return torch.cat(outs)


def fwd_serial():
    return torch.cat([unet(xs[i], t, cs[i], **kw).sample for i in range(2)])


res = {}
with torch.no_grad():
    for name, fn in (("batch 2, one stream", fwd2), ("2 x batch 1, one stream", fwd_serial), ("2 x batch 1, two streams", fwd_split),
                     ("batch 2, one stream", fwd2), ("2 x batch 1, two streams", fwd_split)):
        for _ in range(3):
            y = fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            y = fn()
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            g.replay()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        res[name] = y.float().clone()
        print(f"{name:<28} {dt * 1e3:7.3f} ms per guided forward", flush=True)
a, b = res["batch 2, one stream"], res["2 x batch 1, two streams"]
print("max rel diff split vs batch 2:", float((a - b).abs().max() / a.abs().max()))
