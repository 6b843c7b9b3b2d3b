"""Where do the remaining torch copies / adds of one eager PPFT step come from?  A TorchDispatchMode prints the Python call site of
every aten copy / add / cat / clone on a tensor above 64 KB."""
import os, sys, traceback, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from torch.utils._python_dispatch import TorchDispatchMode
dev = torch.device("cuda", 0)
tr = bench.build(dev, 32)
batch = bench.synthetic_batch(4, dev, 0)
for _ in range(2): tr.step(**batch)
torch.cuda.synchronize()
seen = collections.Counter()
class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__ if hasattr(func, "__name__") else str(func)
        if any(k in str(func) for k in ("copy_", "add.", "add_", "cat", "clone", "_to_copy", "fill_", "zero_", "mul.")):
            big = [a for a in args if isinstance(a, torch.Tensor) and a.numel() * a.element_size() >= 65536]
            if not big and args and isinstance(args[0], (list, tuple)):
                big = [a for a in args[0] if isinstance(a, torch.Tensor) and a.numel() * a.element_size() >= 65536]
            if big:
                st = [f for f in traceback.extract_stack() if "aqualora_amd" in f.filename or f.filename.endswith("bench.py")]
                site = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in st[-3:][::-1])
                seen[(str(func), tuple(big[0].shape), site)] += 1
        return func(*args, **(kwargs or {}))
with Spy():
    tr.step(**batch)
torch.cuda.synchronize()
for (f, shp, site), n in sorted(seen.items(), key=lambda kv: -kv[1] * torch.Size(kv[0][1]).numel()):
    print(f"{n:3d}x {f:28s} {str(shp):24s} {site}")
