"""Experiment: the two forward passes of a PPFT step (frozen 'clean' pass, watermarked LoRA pass) as
  (A) two batch-B forwards on two HIP streams (the round-1 structure), vs
  (B) ONE batch-2B forward on one stream where the clean samples carry an all-zero scale row (bit-identical to scale=None),
  (C) one batch-2B forward with no LoRA at all (lower bound).
All under no_grad inside a HIP graph; prints ms per replay."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from aqualora_amd import ops, synth  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda", 0)
B = int(os.environ.get("B", "4"))
rank = int(os.environ.get("RANK_R", "32"))
tr = bench.build(dev, rank)
unet = tr.unet
b = bench.synthetic_batch(B, dev, 0)
x = b["z"].to(torch.bfloat16)
t, ctx = b["t"], b["ctx"]
S = tr.mapper(b["msg"]).detach()
x2, t2, ctx2 = torch.cat([x, x]), torch.cat([t, t]), torch.cat([ctx, ctx])
S2 = torch.cat([torch.zeros_like(S), S])
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def two_streams():
    main = torch.cuda.current_stream()
    s1.wait_stream(main)
    s2.wait_stream(main)
    with torch.cuda.stream(s1):
        a = unet(x, t, ctx, cross_attention_kwargs={"scale": None}).sample
    with torch.cuda.stream(s2):
        c = unet(x, t, ctx, cross_attention_kwargs={"scale": S}).sample
    main.wait_stream(s1)
    main.wait_stream(s2)
    return a, c


def one_stream_lora():
    return unet(x2, t2, ctx2, cross_attention_kwargs={"scale": S2}).sample


def one_stream_plain():
    return unet(x2, t2, ctx2, cross_attention_kwargs={"scale": None}).sample


def seq_two():
    a = unet(x, t, ctx, cross_attention_kwargs={"scale": None}).sample
    c = unet(x, t, ctx, cross_attention_kwargs={"scale": S}).sample
    return a, c


def timeit(fn, name):
    with torch.no_grad():
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
            fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(10):
            e0.record()
            g.replay()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort()
        print(f"{name:28s} median {ts[len(ts) // 2]:7.3f} ms   min {ts[0]:7.3f} ms", flush=True)
        return out


a, c = timeit(two_streams, f"A two streams B={B}")
o = timeit(one_stream_lora, f"B one stream 2B zero-scale")
timeit(one_stream_plain, f"C one stream 2B no LoRA")
timeit(seq_two, f"D one stream, two passes")
print("clean half identical:", torch.equal(o[:B], a), " wm half identical:", torch.equal(o[B:], c),
      " max diff wm", (o[B:].float() - c.float()).abs().max().item())
