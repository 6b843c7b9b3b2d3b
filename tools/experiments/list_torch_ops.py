"""Which torch-native ops remain in one eager PPFT step (they appear as at::native::* kernels in the step trace)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda", 0)
tr = bench.build(dev, 32)
batch = bench.synthetic_batch(4, dev, 0)
for _ in range(2): tr.step(**batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    tr.step(**batch)
    torch.cuda.synchronize()
rows = {}
for e in prof.events():
    if e.name.startswith("aten::") and e.device_time_total > 0 and e.name not in ("aten::empty", "aten::view"):
        stack = [s for s in (e.stack or []) if ("aqualora_amd" in s or "bench.py" in s) and "_call_impl" not in s]
        key = (e.name, str(e.input_shapes)[:60], " <- ".join(x.split("/")[-1][:48] for x in stack[:3]))
        r = rows.setdefault(key, [0, 0.0]); r[0] += 1; r[1] += e.device_time_total
for (name, shp, st), (n, us) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{us:8.1f} us {n:3d}x {name:18s} {shp:60s} {st}")
