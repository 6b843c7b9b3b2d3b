"""Isolated timing of the twin-shape forward convs (8 x 64 x 64, Cout 320) inside HIP graphs of 100 launches; AQL_CONV_ROW=0/1
selects the implicit-GEMM 256x160 kernel or the row-tile kernel (read once per process)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from aqualora_amd import ops, synth
dev = torch.device("cuda", 0)
NX = int(os.environ.get("NX", "4"))   # rotating inputs: 16 x 21 MB exceeds the 256 MB Infinity Cache
def ev(run, n):
    run(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1e3)
    return sorted(ts)[2]
for B in (8, 4):
  for cin in (320, 640):
    w = synth.normal("k.w", (320, cin, 3, 3), 0.02, 1, dev)
    pk = ops.PackedConv3x3(w, torch.zeros(320, device=dev), 1)
    xs = [synth.normal(f"k.x{i}", (B, cin, 64, 64), 1.0, 1, dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) for i in range(NX)]
    with torch.no_grad():
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(100): ops.conv3x3(xs[i % NX], pk)
        us = ev(g.replay, 100)
    fl = 2.0 * B * 64 * 64 * 320 * 9 * cin
    print(f"B={B} {cin}->320: {us:6.1f} us  {fl / us / 1e6:7.1f} TFLOP/s")
