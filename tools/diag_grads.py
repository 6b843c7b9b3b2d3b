import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.common import *
from tests.test_gpu_parity import _gpu_tiny
from oracle import ppft_oracle as O
from aqualora_amd.ppft import PPFTTrainer
from aqualora_amd.watermark import MapperNet, SecretEncoder
unet, keys, lw = _gpu_tiny()
inp = ppft_inputs(device="cuda")
mapper = MapperNet(48, TINY_RANK)
with torch.no_grad(): mapper.bit_embeddings.weight.copy_(inp["E"])
tr = PPFTTrainer(unet, mapper, SecretEncoder(48, 8, 16), TINY_RANK)
tr.sec_encoder.encode = lambda m, out_scale=1.0: inp["wm"]
loss, pred, clean = tr.forward_backward(inp["z"], inp["msg"], inp["eps"], inp["t"], inp["ctx"])
cpu = ppft_inputs()
for mode in (True, False):
    lo = {k: (d.clone().requires_grad_(True), u.clone().requires_grad_(True)) for k, (d, u) in lw.items()}
    Eo = cpu["E"].clone().requires_grad_(True)
    l, p_o, c_o, _ = O.ppft_loss(tiny_unet().state_dict(), TINY, lo, Eo, cpu["msg"], cpu["z"], cpu["wm"], cpu["eps"], cpu["t"], cpu["ctx"], bf16=mode)
    l.backward()
    print("oracle bf16=%s loss %.6f hip %.6f" % (mode, l.item(), loss.item()))
    rows = []
    for k in keys:
        lay = unet.get_submodule(k).lora_layer
        for nm, got, want in (("down", lay.down.weight.grad, lo[k][0].grad), ("up", lay.up.weight.grad, lo[k][1].grad)):
            got = got.detach().double().cpu().flatten(); want = want.double().flatten()
            rows.append((((got - want).norm() / want.norm()).item(), want.norm().item(), got.norm().item(), k + "." + nm))
    rows.sort(reverse=True)
    for r in rows[:12]: print("  %.3f  ref|g|=%.3e hip|g|=%.3e %s" % r)
    print("  median l2rel %.3f" % np.median([r[0] for r in rows]))
    import collections
    by = collections.defaultdict(list)
    for r in rows: by[r[3].split(".")[-2] + "." + r[3].split(".")[-1] if "to_out" not in r[3] else "to_out." + r[3].split(".")[-1]].append(r[0])
    print("  by type:", {k: round(float(np.mean(v)), 3) for k, v in by.items()})
