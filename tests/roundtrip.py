"""The watermark round trip on the HIP path, end to end, for enough steps to show that training trains:

    stage 1   SecretEncoder + SecretDecoder trained jointly through the frozen VAE decoder     train/latent_wm_pretrain.py:159-225
              (held-out accuracy in eval mode like :232-240)
    PPFT      watermark-LoRA + MapperNet trained on the frozen U-Net, captured step graph       train/ppft_train.py:987-1068
    bake      down' = diag(S(m)) . down . 1.03                                                  scripts/create_wm_lora.py:24-41
    fuse      W += up . down'                                                                   evaluation/utils_eval.py:81-82
    sample    DDIM + classifier-free guidance 7.5 on the fused U-Net                            utils_eval.py:83-126
    extract   VAE decode -> SecretDecoder -> argmax -> bit accuracy                             utils_eval.py:131-140,156-213
              (the reference's own in-script check of a run: ppft_train.py:1170-1183)

A toy world, sized for a test (about 2.5 minutes on one MI355X + its host): tests/common.TINY U-Net on 16x16x4 latents, a
reduced-width VAE (128x128 images), the full-size EfficientNet-B1 decoder at its fixed 512x512 (utils/models.py:92-94), 16-bit
messages on a rank-32 LoRA (the one-launch LoRA kernels; 16 bits instead of 48 because a from-scratch decoder has minutes, not
epochs).  Three things the real recipe gets from pre-trained weights have to be made here:
  * the frozen U-Net must be a (rough) denoiser, or sampling from it means nothing (a randomly initialised one blows DDIM samples
    up to std ~10): `pretrain_unet_cpu` gives it 300 Adam steps of epsilon prediction on smooth random fields, on the CPU through the
    oracle's U-Net -- the HIP path has no weight gradient for the frozen base, by design;
  * the data set is then what that U-Net itself samples (`latent_pool`): SD-1.5's samples are distributed like its training data,
    a 300-step toy's are not, so the world is made self-consistent the other way round;
  * the decoder starts from He-normal weights (no ImageNet), the encoder's conv from N(0, 2^2) instead of zeros, and the first 100
    steps use a constant cover like the reference's --fixinit (latent_wm_pretrain.py:165-167).
A quarter of the PPFT samples carry the unconditional text state (zeros): classifier-free guidance evaluates the watermarked
U-Net on it at every sampling step, so the prior has to be preserved there too (without it: bit accuracy 0.55 at guidance 7.5
against 0.82 with it, measured).  Every random draw is counter-based (aqualora_amd.synth), so the oracle restatement
(oracle/roundtrip_oracle.py) sees the same batches.  Test infrastructure (it pre-trains through oracle/): tests only.

    python -m tests.roundtrip ['{"ppft_steps": 400, ...}']      on an MI355X: prints the trajectories
"""
import json
import os
import sys
import time

import torch

from aqualora_amd import synth

VAE_SCALING = 0.18215
RES = 16
SEED = 4242
TINY_VAE_CH = (32, 64, 64, 64)


def default_cfg():
    return dict(bits=16, rank=32, pool=256, unet_steps=300, unet_batch=16,
                stage1_steps=500, stage1_batch=16, stage1_lr=1e-3, stage1_wd=1e-4, stage1_fixinit=100, enc_conv_std=2.0,
                ppft_steps=1500, ppft_batch=16, ppft_lr=2e-3, sample_steps=20, guidance=7.5, eval_images=16)


# ------------------------------------------------------------------------------------------------ the toy world
def text_states(tag, n, dev, dim=32):
    return synth.normal(tag, (n, 77, dim), 1.0, SEED, dev)


def toy_fields(tag, n, dev="cpu"):
    """Smooth random fields (a 4x4 Gaussian grid, bilinearly upsampled) plus a little white noise, about unit variance."""
    lo = synth.normal(tag + ".lo", (n, 4, 4, 4), 1.0, SEED)
    hi = synth.normal(tag + ".hi", (n, 4, RES, RES), 1.0, SEED)
    z = torch.nn.functional.interpolate(lo, size=RES, mode="bilinear", align_corners=False) * 1.2 + 0.3 * hi
    return z.to(dev).contiguous()


def pretrain_unet_cpu(steps, batch, log=None):
    """Adam steps of the epsilon-prediction loss on `toy_fields`, on the CPU through the oracle's U-Net (see the module docstring).
    A quarter of the batches see the unconditional (zero) text state.  Returns an fp32 state dict."""
    from oracle import ppft_oracle as O
    from tests.common import TINY, tiny_unet
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in tiny_unet().state_dict().items()}
    acp = O.alphas_cumprod()
    opt = torch.optim.Adam(list(sd.values()), lr=1e-3)
    for i in range(steps):
        z = toy_fields(f"rt.unet.z{i}", batch)
        eps = synth.normal(f"rt.unet.eps{i}", (batch, 4, RES, RES), 1.0, SEED)
        t = synth.randint(f"rt.unet.t{i}", (batch,), 1000, SEED)
        ctx = text_states(f"rt.unet.ctx{i}", batch, "cpu")
        if i % 4 == 0:
            ctx = torch.zeros_like(ctx)
        pred = O.UNetOracle(sd, dict(TINY)).forward(O.add_noise(z, eps, t, acp), t, ctx, None)
        loss = ((pred - eps) ** 2).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
        if log and (i % 50 == 0 or i == steps - 1):
            log(f"unet   {i:4d} eps-loss {float(loss.detach()):.4f}")
    return {k: v.detach() for k, v in sd.items()}


def vae_cfg():
    from aqualora_amd.vae import SD15_VAE
    return dict(SD15_VAE, block_out_channels=TINY_VAE_CH)


def vae_state():
    from aqualora_amd.vae import synthetic_state_dict
    return synthetic_state_dict(vae_cfg(), seed=SEED)


def make_unet(dev, unet_sd):
    from tests.common import tiny_unet
    unet = tiny_unet(dev, torch.bfloat16)
    unet.load_state_dict({k: v.to(dev) for k, v in unet_sd.items()})
    return unet


def make_vae(dev):
    from aqualora_amd.vae import AutoencoderKL
    return AutoencoderKL(vae_state(), vae_cfg(), dev)


def latent_pool(unet, n, steps, dev):
    """`n` latents sampled by the plain frozen U-Net itself (DDIM, no guidance): the data set stage 1 and PPFT train on."""
    from aqualora_amd.inference import ddim_sample
    out = []
    for i in range(0, n, 32):
        m = min(32, n - i)
        ctx = text_states(f"rt.pool.ctx{i}", m, dev)
        x = synth.normal(f"rt.pool.x{i}", (m, 4, RES, RES), 1.0, SEED, dev)
        out.append(ddim_sample(unet, ctx, torch.zeros_like(ctx), x, steps, 1.0, graph=False))
    return torch.cat(out).contiguous()


# ------------------------------------------------------------------------------------------------ initial states (synth: any device)
def encoder_init(cfg):
    """{lin_w, lin_b, conv_w, conv_b} of SecretEncoder(bits, base_res 8, resolution 16)."""
    bits, R = cfg["bits"], RES // 2
    return dict(lin_w=synth.normal("rt.enc.lin.w", (R * R, bits), bits ** -0.5, SEED), lin_b=torch.zeros(R * R),
                conv_w=synth.normal("rt.enc.conv.w", (4, 4, 3, 3), cfg["enc_conv_std"], SEED), conv_b=torch.zeros(4))


def make_encoder(cfg, dev, state=None):
    from aqualora_amd.watermark import SecretEncoder
    st = state or encoder_init(cfg)
    enc = SecretEncoder(cfg["bits"], base_res=RES // 2, resolution=RES)
    with torch.no_grad():
        enc.secret_scaler[0].weight.copy_(st["lin_w"])
        enc.secret_scaler[0].bias.copy_(st["lin_b"])
        enc.secret_scaler[5].weight.copy_(st["conv_w"])
        enc.secret_scaler[5].bias.copy_(st["conv_b"])
    return enc.to(dev)


def encoder_state(enc):
    lin, conv = enc.secret_scaler[0], enc.secret_scaler[5]
    return {k: v.detach().float().cpu().clone() for k, v in
            dict(lin_w=lin.weight, lin_b=lin.bias, conv_w=conv.weight, conv_b=conv.bias).items()}


def make_decoder(cfg, dev):
    """He-normal convolutions, BatchNorm gamma 1 / beta 0 (the reference starts from ImageNet weights, which do not exist here)."""
    from aqualora_amd.decoder import SecretDecoder
    dec = SecretDecoder(cfg["bits"])
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
                t.copy_(synth.normal("rt.dec." + name, tuple(t.shape), (2.0 / t[0].numel()) ** 0.5, SEED))
    return dec.to(dev)


def lora_init(unet, cfg):
    """diffusers' init: down ~ N(0, (1/r)^2), up = 0 -> {site: (down, up)} on the CPU."""
    from aqualora_amd.unet import lora_keys
    r, out = cfg["rank"], {}
    for k in lora_keys(unet):
        m = unet.get_submodule(k)
        if hasattr(m, "in_channels"):
            ds, us = (r, m.in_channels, 1, 1), (m.out_channels, r, 1, 1)
        else:
            ds, us = (r, m.in_features), (m.out_features, r)
        out[k] = (synth.normal(k + ".rt.down", ds, 1.0 / r, SEED), torch.zeros(us))
    return out


def mapper_init(cfg):
    return synth.normal("rt.mapper.E", (cfg["bits"], cfg["rank"]), 1.0, SEED)


# ------------------------------------------------------------------------------------------------ batches (synth: any device)
def stage1_batch(i, pool, cfg, tag="rt.s1"):
    """Step i of stage 1: raw latents (latent_wm_pretrain.py:171 decodes unscaled latents), message, and the decoder's train-mode
    draws -- stochastic-depth survival factors per block ([23][B], torchvision's 0.2 * block / 23 schedule) and the dropout mask."""
    B, dev = cfg["stage1_batch"], pool.device
    idx = synth.randint(f"{tag}.idx{i}", (B,), pool.shape[0], SEED).tolist()
    lat = pool[idx] / VAE_SCALING
    if i < cfg["stage1_fixinit"]:
        lat = torch.zeros_like(lat)
    nblk = 23
    sd_noise = []
    for b in range(nblk):
        p = 0.2 * b / nblk
        keep = (synth.randint(f"{tag}.sd{i}.{b}", (B,), 1 << 20, SEED, dev).float() / float(1 << 20)) >= p
        sd_noise.append(keep.float() / (1.0 - p))
    drop = ((synth.randint(f"{tag}.drop{i}", (B, 1280), 1 << 20, SEED, dev).float() / float(1 << 20)) >= 0.2).float() / 0.8
    return dict(lat=lat.contiguous(), msg=synth.bits(f"{tag}.msg{i}", (B, cfg["bits"]), SEED, dev), sd_noise=sd_noise, drop_mask=drop)


def ppft_batch(i, pool, cfg, tag="rt.pp"):
    B, dev = cfg["ppft_batch"], pool.device
    idx = synth.randint(f"{tag}.idx{i}", (B,), pool.shape[0], SEED).tolist()
    ctx = text_states(f"{tag}.ctx{i}", B, dev)
    ctx[synth.randint(f"{tag}.cfgdrop{i}", (B,), 4, SEED, dev) == 0] = 0      # the unconditional text state, see the module docstring
    return dict(z=pool[idx].contiguous(), msg=synth.bits(f"{tag}.msg{i}", (B, cfg["bits"]), SEED, dev),
                eps=synth.normal(f"{tag}.eps{i}", (B, 4, RES, RES), 1.0, SEED, dev),
                t=synth.randint(f"{tag}.t{i}", (B,), 1000, SEED, dev), ctx=ctx)


# ------------------------------------------------------------------------------------------------ the HIP recipe
class _DecoderWithDraws:
    """Stage1Step calls ``sec_decoder(image)``; this hands the step's stochastic-depth / dropout draws to the train-mode forward."""

    def __init__(self, dec):
        self.dec, self.draws = dec, {}

    def __call__(self, x):
        return self.dec(x, **self.draws)


def stage1(vae, pool, cfg, dev, steps=None, enc=None, dec=None, log=None, grads=None):
    """latent_wm_pretrain.py:159-225 in its message-loss phase (loss = msgloss, :207-213 at epoch <= 6), watermark at scale 1.0 (past
    the warm-up, :174-177), Identity noiser, AdamW(lr, weight_decay 1e-4) on encoder + decoder (:125-128).
    Returns (encoder, decoder, [(msgloss, train accuracy)])."""
    from aqualora_amd import noise as NZ, stage1 as S1
    enc = enc if enc is not None else make_encoder(cfg, dev)
    dec = (dec if dec is not None else make_decoder(cfg, dev)).train()
    opt = torch.optim.AdamW(list(enc.parameters()) + list(dec.parameters()), lr=cfg["stage1_lr"], weight_decay=cfg["stage1_wd"])
    dd = _DecoderWithDraws(dec)
    step = S1.Stage1Step(enc, dd, lambda z: vae.decode_grad(z, scaled=False), NZ.Noiser(["Identity"], [1.0]))
    step.warmup = False
    traj = []
    n = cfg["stage1_steps"] if steps is None else steps
    for i in range(n):
        b = stage1_batch(i, pool, cfg)
        dd.draws = dict(sd_noise=b["sd_noise"], drop_mask=b["drop_mask"])
        opt.zero_grad(set_to_none=True)
        out = step.losses(b["lat"], b["msg"], epochs_done=0, combine=dict(cornerfy_aug=False), noiser_choice=[1.0])
        out["loss"].backward()
        if grads is not None:      # the encoder's gradients of this step, as backward left them (tests)
            lin, conv = enc.secret_scaler[0], enc.secret_scaler[5]
            grads.append({k: v.grad.detach().float().cpu().clone() for k, v in
                          dict(lin_w=lin.weight, lin_b=lin.bias, conv_w=conv.weight, conv_b=conv.bias).items()})
        opt.step()
        acc = float((out["logits"].argmax(-1) == b["msg"].long()).float().mean())
        traj.append((float(out["msgloss"].detach()), acc))
        if log and (i % 50 == 0 or i == n - 1):
            log(f"stage1 {i:4d} msgloss {traj[-1][0]:.4f} train-acc {acc:.3f}")
    return enc, dec, traj


@torch.no_grad()
def stage1_heldout_accuracy(enc, dec, vae, pool, cfg, n=32):
    """latent_wm_pretrain.py:232-240: eval mode, fresh messages, decode(latents + wm) -> decoder -> 1 - BER."""
    dec.eval()
    lat = pool[-n:] / VAE_SCALING
    msg = synth.bits("rt.s1.val.msg", (n, cfg["bits"]), SEED + 1, pool.device)
    wm_lat, _ = enc(lat, msg.float())
    img = vae.decode(wm_lat, scaled=False)
    return float((dec(img).argmax(-1) == msg.long()).float().mean())


def make_trainer(unet, enc, cfg, dev, lora=None, E=None):
    from aqualora_amd.lora import inject_lora
    from aqualora_amd.ppft import PPFTTrainer
    from aqualora_amd.unet import lora_keys
    from aqualora_amd.watermark import MapperNet
    r = cfg["rank"]
    keys = lora_keys(unet)
    inject_lora(unet, r, keys)
    lora = lora if lora is not None else lora_init(unet, cfg)
    with torch.no_grad():
        for k in keys:
            lay = unet.get_submodule(k).lora_layer
            lay.down.weight.copy_(lora[k][0])
            lay.up.weight.copy_(lora[k][1])
    mapper = MapperNet(cfg["bits"], r)
    with torch.no_grad():
        mapper.bit_embeddings.weight.copy_(E if E is not None else mapper_init(cfg))
    return PPFTTrainer(unet, mapper, enc, r, learning_rate=cfg["ppft_lr"])


def ppft(tr, pool, cfg, steps=None, graph=True, log=None, first=0):
    """ppft_train.py:987-1068 for `steps` steps (batches first, first+1, ...) on the captured step graph (or eagerly)."""
    n = cfg["ppft_steps"] if steps is None else steps

    def batch(i):
        b = ppft_batch(i, pool, cfg)
        b["ctx"] = b["ctx"].to(torch.bfloat16)
        return b

    run = tr.capture(batch(first), warmup=0) if graph else tr.step
    traj = []
    for i in range(first, first + n):
        traj.append(float(run(**batch(i))))
        if log and (i % 100 == 0 or i == first + n - 1):
            log(f"ppft   {i:4d} loss {traj[-1]:.6f}")
    return traj


@torch.no_grad()
def sample_and_extract(unet_factory, tr, dec, vae, cfg, dev, log=None):
    """create_watermark_lora -> fuse_lora -> DDIM + CFG -> VAE decode -> SecretDecoder -> bit accuracy for `eval_images` held-out
    messages, one baked + fused U-Net per message (one LoRA file per user in the reference).  Controls: the same pipeline on the
    plain U-Net (no watermark: chance), and the plain sample + the encoder's residual (what a perfect PPFT would produce,
    ppft_train.py:994-996,1011); `shift_cosine` = how well the latent shift the LoRA produced lines up with that residual."""
    from aqualora_amd.checkpoint import lora_state_dict
    from aqualora_amd.inference import create_watermark_lora, ddim_sample, fuse_lora
    from aqualora_amd.lora import inject_lora
    from aqualora_amd.unet import lora_keys
    dec.eval()
    sd = lora_state_dict(tr.unet, lora_keys(tr.unet))
    n = cfg["eval_images"]
    msgs = synth.bits("rt.eval.msg", (n, cfg["bits"]), SEED + 2, dev)
    acc, acc_plain, acc_ideal, cos = [], [], [], []
    plain = unet_factory()
    for i in range(n):
        m = msgs[i:i + 1]
        _, baked = create_watermark_lora(sd, tr.mapper, m.cpu())
        fused = unet_factory()
        inject_lora(fused, cfg["rank"], lora_keys(fused))
        fuse_lora(fused, baked, 1.0, lora_keys(fused))
        ctx = text_states(f"rt.eval.ctx{i}", 1, dev)
        x = synth.normal(f"rt.eval.x{i}", (1, 4, RES, RES), 1.0, SEED, dev)
        z_w = ddim_sample(fused, ctx, torch.zeros_like(ctx), x, cfg["sample_steps"], cfg["guidance"], graph=False)
        z_p = ddim_sample(plain, ctx, torch.zeros_like(ctx), x, cfg["sample_steps"], cfg["guidance"], graph=False)
        wm = tr.sec_encoder.encode(m.float(), out_scale=VAE_SCALING)
        for z, dst in ((z_w, acc), (z_p, acc_plain), (z_p + wm, acc_ideal)):
            bits = dec(vae.decode(z / VAE_SCALING, scaled=False)).argmax(-1)
            dst.append(float((bits == m.long()).float().mean()))
        d = (z_w - z_p).flatten()
        cos.append(float(torch.dot(d, wm.flatten()) / (d.norm() * wm.norm() + 1e-30)))
    res = dict(bit_accuracy=sum(acc) / n, bit_accuracy_plain=sum(acc_plain) / n, bit_accuracy_ideal=sum(acc_ideal) / n,
               per_image=acc, shift_cosine=sum(cos) / n)
    if log:
        log("extract: " + json.dumps({k: round(v, 3) for k, v in res.items() if k != "per_image"}))
    return res


def recipe(cfg=None, dev="cuda", log=None, graph=True):
    cfg = dict(default_cfg(), **(cfg or {}))
    t0 = time.time()
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    unet_sd = pretrain_unet_cpu(cfg["unet_steps"], cfg["unet_batch"], log)
    t_unet = time.time() - t0
    unet, vae = make_unet(dev, unet_sd), make_vae(dev)
    pool = latent_pool(unet, cfg["pool"], cfg["sample_steps"], dev)
    if log:
        log(f"pool: {tuple(pool.shape)} std {float(pool.std()):.3f} absmax {float(pool.abs().max()):.2f}  ({time.time() - t0:.1f} s)")
    enc, dec, s1 = stage1(vae, pool, cfg, dev, log=log)
    s1_acc = stage1_heldout_accuracy(enc, dec, vae, pool, cfg)
    if log:
        log(f"stage1 held-out accuracy {s1_acc:.3f}  ({time.time() - t0:.1f} s)")
    tr = make_trainer(unet, enc, cfg, dev)
    pp = ppft(tr, pool, cfg, graph=graph, log=log)
    res = sample_and_extract(lambda: make_unet(dev, unet_sd), tr, dec, vae, cfg, dev, log)
    res.update(stage1=s1, stage1_heldout_accuracy=s1_acc, ppft=pp, seconds=time.time() - t0, seconds_unet_pretrain_cpu=t_unet,
               pool_std=float(pool.std()))
    res["_state"] = dict(unet_sd=unet_sd, pool=pool, enc=enc, dec=dec, vae=vae, trainer=tr, cfg=cfg)
    return res


if __name__ == "__main__":
    over = json.loads(sys.argv[1]) if len(sys.argv) > 1 else {}
    out = recipe(over, log=lambda s: print(s, flush=True))
    print(json.dumps({k: v for k, v in out.items() if k not in ("stage1", "ppft", "_state")}))
