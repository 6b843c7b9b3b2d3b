"""Exploration helper for tests/roundtrip.py: one U-Net pre-training + one stage 1, then several PPFT / sampling variants."""
import json, sys, time, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import roundtrip as R
log = lambda s: print(s, flush=True)
base = dict(R.default_cfg(), stage1_steps=500, stage1_fixinit=100, enc_conv_std=2.0, stage1_batch=16, eval_images=16)
base.update(json.loads(sys.argv[1]) if len(sys.argv) > 1 else {})
variants = json.loads(sys.argv[2]) if len(sys.argv) > 2 else [{}]
dev = "cuda"
torch.set_num_threads(16)
unet_sd = R.pretrain_unet_cpu(base["unet_steps"], base["unet_batch"], None)
unet, vae = R.frozen_models(dev, unet_sd)
pool = R.latent_pool(unet, base["pool"], base["sample_steps"], dev)
log(f"pool std {float(pool.std()):.3f}")
enc, dec, s1 = R.stage1(vae, pool, base, dev, None)
log(f"stage1 final loss {s1[-1][0]:.3f} held-out {R.stage1_heldout_accuracy(enc, dec, vae, pool, dev):.3f}")
for v in variants:
    cfg = dict(base, **v)
    t0 = time.time()
    dec.train()
    u = R.make_unet(dev, unet_sd)
    tr, pp = R.ppft(u, enc, pool, cfg, dev, None)
    log(f"== {v}: ppft loss first10 {sum(pp[:10]) / 10:.4f} last50 {sum(pp[-50:]) / 50:.4f}")
    R.sample_and_extract(lambda: R.make_unet(dev, unet_sd), tr, dec, vae, cfg, dev, log)
    log(f"   ({time.time() - t0:.1f} s)")
