"""The 8x8-map convolutions (weight-streaming-bound: 29.5 / 59 MB of weights for 15 / 30 GFLOP at 8 samples): tile / split sweep
with HBM-cold weights (12 weight sets per graph, 354+ MB).  AQL_TILE / AQL_SPLITS are re-read per call."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from aqualora_amd import ops, synth
dev = torch.device("cuda", 0)
def gt(run, n):
    run(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1e3)
    return sorted(ts)[2]
for (B, cin, cout, hw) in ((8, 1280, 1280, 8), (8, 2560, 1280, 8), (4, 1280, 1280, 8)):
    NW = int(os.environ.get('NW', '12'))
    pks = [ops.PackedConv3x3(synth.normal(f"w{i}", (cout, cin, 3, 3), 0.02, 1, dev), torch.zeros(cout, device=dev), 1) for i in range(NW)]
    x = synth.normal("x", (B, cin, hw, hw), 1.0, 1, dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    line = f"B={B} {cin}->{cout} @{hw}x{hw}:"
    for tile in ("",):
        for sp in ("", "4", "6", "8", "12", "16"):
            for k, v in (("AQL_TILE", tile), ("AQL_SPLITS", sp)):
                if v: os.environ[k] = v
                else: os.environ.pop(k, None)
            try:
                with torch.no_grad():
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        for i in range(12): ops.conv3x3(x, pks[i % NW])
                    us = gt(g.replay, 12)
                line += f"  t{tile or 'auto'}/s{sp or 'auto'}={us:.1f}"
            except Exception as e:
                line += f"  t{tile}/s{sp}=ERR"
    print(line, flush=True)
