"""The watermark round trip on the HIP path, end to end, for enough steps to show that training trains:

    stage 1   SecretEncoder + SecretDecoder trained jointly through the frozen VAE decoder     train/latent_wm_pretrain.py:159-225
              (per-epoch held-out accuracy like :232-240)
    PPFT      watermark-LoRA + MapperNet trained on the frozen U-Net, captured step graph       train/ppft_train.py:987-1068
    bake      down' = diag(S(m)) . down . 1.03                                                  scripts/create_wm_lora.py:24-41
    fuse      W += up . down'                                                                   evaluation/utils_eval.py:81-82
    sample    DDIM + classifier-free guidance on the fused U-Net                                utils_eval.py:83-126
    extract   VAE decode -> SecretDecoder -> argmax -> bit accuracy                             utils_eval.py:131-140,156-213
              (the reference's own in-script check: ppft_train.py:1170-1183)

Everything is synthetic and tiny (tests/common.TINY U-Net, 16x16x4 latents, a reduced-width VAE: 128x128 images; the
EfficientNet-B1 decoder is full size and always works at 512x512, utils/models.py:92-94), the weights of the frozen parts are
the counter-based synthetic ones.  The "dataset" of latents is what the plain U-Net itself samples (the role real-image latents
play for SD-1.5), drawn once up front.  `recipe()` returns every loss trajectory and accuracy; tests/test_roundtrip.py asserts
on them and compares the first steps with the same recipe in oracle/roundtrip_oracle.py.  Test infrastructure: imports
oracle-free product code only.

    python -m tests.roundtrip            (on an MI355X: prints the trajectories)
"""
import json
import sys
import time

import torch

from aqualora_amd import synth

VAE_SCALING = 0.18215
BITS = 48
RES = 16
SEED = 4242


def default_cfg():
    return dict(pool=256, pool_steps=10, stage1_steps=300, stage1_batch=8, stage1_lr=1e-3, ppft_steps=300, ppft_batch=8,
                ppft_lr=1e-3, rank=8, sample_steps=20, guidance=3.0, eval_images=8, lat_gain=1.0)


def frozen_models(dev):
    """Plain tiny U-Net (no LoRA yet), reduced-width VAE, both with the synthetic frozen weights."""
    from aqualora_amd.vae import SD15_VAE, AutoencoderKL, synthetic_state_dict
    from tests.common import tiny_unet
    tiny_vae = dict(SD15_VAE, block_out_channels=(32, 64, 64, 64))
    unet = tiny_unet(dev, torch.bfloat16)
    vae = AutoencoderKL(synthetic_state_dict(tiny_vae, seed=SEED), tiny_vae, dev)
    return unet, vae


def text_states(tag, n, dev, dim=32):
    return synth.normal(tag, (n, 77, dim), 1.0, SEED, dev)


def latent_pool(unet, n, steps, dev, gain=1.0):
    """`n` scaled latents sampled by the plain U-Net (DDIM, guidance 1): the data distribution of this toy world."""
    from aqualora_amd.inference import ddim_sample
    out = []
    for i in range(0, n, 32):
        m = min(32, n - i)
        ctx = text_states(f"rt.pool.ctx{i}", m, dev)
        x = synth.normal(f"rt.pool.x{i}", (m, 4, RES, RES), 1.0, SEED, dev)
        out.append(ddim_sample(unet, ctx, torch.zeros_like(ctx), x, steps, 1.0, graph=False))
    return (torch.cat(out) * gain).contiguous()


def init_decoder(dec):
    """He-normal convolutions, BatchNorm gamma 1 / beta 0 (the reference starts from ImageNet weights, which do not exist here)."""
    with torch.no_grad():
        for name, t in list(dec.named_parameters()) + list(dec.named_buffers()):
            if name.endswith("running_var"):
                t.fill_(1.0)
            elif name.endswith("running_mean") or name.endswith("num_batches_tracked"):
                t.zero_()
            elif name.endswith(".1.weight") and t.dim() == 1:
                t.fill_(1.0)
            elif t.dim() == 1:
                t.zero_()
            else:
                fan = t[0].numel()
                t.copy_(synth.normal("rt.dec." + name, tuple(t.shape), (2.0 / fan) ** 0.5, SEED))
    return dec


def stage1(vae, pool, cfg, dev, log=None):
    """latent_wm_pretrain.py:159-225 in its message-loss phase (epochs <= 6: loss = msgloss), scale 1.0 (post warm-up), Identity /
    JPEG noiser, AdamW on encoder + decoder; raw (unscaled) latents like :171.  Returns (encoder, decoder, trajectory)."""
    from aqualora_amd import noise as NZ, stage1 as S1
    from aqualora_amd.decoder import SecretDecoder
    from aqualora_amd.watermark import SecretEncoder
    enc = SecretEncoder(BITS, base_res=RES // 2, resolution=RES)
    with torch.no_grad():
        lin = enc.secret_scaler[0]
        lin.weight.copy_(synth.normal("rt.enc.lin.w", tuple(lin.weight.shape), BITS ** -0.5, SEED))
        lin.bias.zero_()
        # the reference zero-initialises the conv (utils/models.py:63-66) and escapes that fixed point over thousands of steps;
        # the toy recipe starts from a small non-zero conv instead
        enc.secret_scaler[5].weight.copy_(synth.normal("rt.enc.conv.w", (4, 4, 3, 3), 0.2, SEED))
    enc = enc.to(dev)
    dec = init_decoder(SecretDecoder(BITS)).to(dev).train()
    opt = torch.optim.AdamW(list(enc.parameters()) + list(dec.parameters()), lr=cfg["stage1_lr"])
    step = S1.Stage1Step(enc, dec, lambda z: vae.decode_grad(z, scaled=False), NZ.Noiser(["Identity"], [1.0]))
    step.warmup = False
    B = cfg["stage1_batch"]
    traj = []
    for i in range(cfg["stage1_steps"]):
        idx = synth.randint(f"rt.s1.idx{i}", (B,), pool.shape[0], SEED).tolist()
        lat = pool[idx] / VAE_SCALING
        msg = synth.bits(f"rt.s1.msg{i}", (B, BITS), SEED, dev)
        opt.zero_grad(set_to_none=True)
        out = step.losses(lat, msg, epochs_done=0, combine=dict(cornerfy_aug=False), noiser_choice=[1.0])
        out["loss"].backward()
        opt.step()
        acc = float((out["logits"].argmax(-1) == msg.long()).float().mean())
        traj.append((float(out["msgloss"]), acc))
        if log and (i % 25 == 0 or i == cfg["stage1_steps"] - 1):
            log(f"stage1 {i:4d} msgloss {traj[-1][0]:.4f} train-acc {acc:.3f}")
    return enc, dec, traj


@torch.no_grad()
def stage1_heldout_accuracy(enc, dec, vae, pool, dev, n=16):
    """latent_wm_pretrain.py:232-240: eval mode, fresh messages, decode(latents + wm) -> decoder -> 1 - BER."""
    dec.eval()
    lat = pool[-n:] / VAE_SCALING
    msg = synth.bits("rt.s1.val.msg", (n, BITS), SEED + 1, dev)
    wm_lat, _ = enc(lat, msg.float())
    img = vae.decode(wm_lat, scaled=False)
    acc = float((dec(img).argmax(-1) == msg.long()).float().mean())
    return acc


def ppft(unet, enc, pool, cfg, dev, log=None, graph=True):
    """ppft_train.py:987-1068 for `ppft_steps` steps on the captured step graph (or eagerly): fresh latents / messages / noise /
    timesteps / text states every step.  Returns (trainer, loss trajectory)."""
    from aqualora_amd.lora import inject_lora
    from aqualora_amd.ppft import PPFTTrainer
    from aqualora_amd.unet import lora_keys
    from aqualora_amd.watermark import MapperNet
    r = cfg["rank"]
    keys = lora_keys(unet)
    inject_lora(unet, r, keys)
    with torch.no_grad():   # diffusers' init: down ~ N(0, 1/r), up = 0
        for k in keys:
            lay = unet.get_submodule(k).lora_layer
            lay.down.weight.copy_(synth.normal(k + ".rt.down", tuple(lay.down.weight.shape), 1.0 / r, SEED, dev))
            lay.up.weight.zero_()
    mapper = MapperNet(BITS, r)
    with torch.no_grad():
        mapper.bit_embeddings.weight.copy_(synth.normal("rt.mapper.E", (BITS, r), 1.0, SEED))
    tr = PPFTTrainer(unet, mapper, enc, r, learning_rate=cfg["ppft_lr"])
    B = cfg["ppft_batch"]

    def batch(i):
        idx = synth.randint(f"rt.pp.idx{i}", (B,), pool.shape[0], SEED).tolist()
        return dict(z=pool[idx].contiguous(), msg=synth.bits(f"rt.pp.msg{i}", (B, BITS), SEED, dev),
                    eps=synth.normal(f"rt.pp.eps{i}", (B, 4, RES, RES), 1.0, SEED, dev),
                    t=synth.randint(f"rt.pp.t{i}", (B,), 1000, SEED, dev),
                    ctx=text_states(f"rt.pp.ctx{i}", B, dev).to(torch.bfloat16))

    run = tr.capture(batch(0), warmup=0) if graph else tr.step
    traj = []
    for i in range(cfg["ppft_steps"]):
        loss = run(**batch(i))
        traj.append(float(loss))
        if log and (i % 25 == 0 or i == cfg["ppft_steps"] - 1):
            log(f"ppft   {i:4d} loss {traj[-1]:.6f}")
    return tr, traj


@torch.no_grad()
def sample_and_extract(unet_factory, tr, dec, vae, cfg, dev, log=None):
    """create_watermark_lora -> fuse_lora -> DDIM(+CFG) -> VAE decode -> SecretDecoder -> bit accuracy, for `eval_images`
    held-out messages (one fused U-Net per message, like one baked LoRA file per user in the reference); also the accuracy
    of the SAME pipeline on the plain U-Net (no watermark: chance level) and the latent shift the LoRA produced."""
    from aqualora_amd.checkpoint import lora_state_dict
    from aqualora_amd.inference import create_watermark_lora, ddim_sample, fuse_lora
    from aqualora_amd.unet import lora_keys
    dec.eval()
    sd = lora_state_dict(tr.unet, lora_keys(tr.unet))
    n = cfg["eval_images"]
    msgs = synth.bits("rt.eval.msg", (n, BITS), SEED + 2, dev)
    accs, accs_plain, shift = [], [], []
    plain = unet_factory()
    for i in range(n):
        bits_str, baked = create_watermark_lora(sd, tr.mapper, msgs[i:i + 1].cpu())
        fused = unet_factory()
        from aqualora_amd.lora import inject_lora
        inject_lora(fused, cfg["rank"], lora_keys(fused))
        fuse_lora(fused, baked, 1.0, lora_keys(fused))
        ctx = text_states(f"rt.eval.ctx{i}", 1, dev)
        x = synth.normal(f"rt.eval.x{i}", (1, 4, RES, RES), 1.0, SEED, dev)
        z_w = ddim_sample(fused, ctx, torch.zeros_like(ctx), x, cfg["sample_steps"], cfg["guidance"], graph=False)
        z_p = ddim_sample(plain, ctx, torch.zeros_like(ctx), x, cfg["sample_steps"], cfg["guidance"], graph=False)
        img_w = vae.decode(z_w / VAE_SCALING, scaled=False)
        img_p = vae.decode(z_p / VAE_SCALING, scaled=False)
        bw = dec(img_w).argmax(-1)
        bp = dec(img_p).argmax(-1)
        accs.append(float((bw == msgs[i:i + 1].long()).float().mean()))
        accs_plain.append(float((bp == msgs[i:i + 1].long()).float().mean()))
        shift.append(float((z_w - z_p).norm() / z_p.norm()))
    if log:
        log(f"extract: watermarked {sum(accs) / n:.3f}  plain {sum(accs_plain) / n:.3f}  latent shift {sum(shift) / n:.3f}")
    return dict(bit_accuracy=sum(accs) / n, bit_accuracy_plain=sum(accs_plain) / n, per_image=accs, latent_shift=sum(shift) / n)


def recipe(cfg=None, dev="cuda", log=None, graph=True):
    cfg = dict(default_cfg(), **(cfg or {}))
    t0 = time.time()
    unet, vae = frozen_models(dev)
    pool = latent_pool(unet, cfg["pool"], cfg["pool_steps"], dev, cfg["lat_gain"])
    if log:
        log(f"pool: {tuple(pool.shape)} std {float(pool.std()):.3f} absmax {float(pool.abs().max()):.2f}  ({time.time() - t0:.1f} s)")
    enc, dec, s1 = stage1(vae, pool, cfg, dev, log)
    s1_acc = stage1_heldout_accuracy(enc, dec, vae, pool, dev)
    if log:
        log(f"stage1 held-out accuracy {s1_acc:.3f}  ({time.time() - t0:.1f} s)")
    tr, pp = ppft(unet, enc, pool, cfg, dev, log, graph)
    from tests.common import tiny_unet
    res = sample_and_extract(lambda: tiny_unet(dev, torch.bfloat16), tr, dec, vae, cfg, dev, log)
    res.update(stage1=s1, stage1_heldout_accuracy=s1_acc, ppft=pp, seconds=time.time() - t0)
    return res


if __name__ == "__main__":
    over = json.loads(sys.argv[1]) if len(sys.argv) > 1 else {}
    out = recipe(over, log=lambda s: print(s, flush=True))
    print(json.dumps({k: v for k, v in out.items() if k not in ("stage1", "ppft")}))
