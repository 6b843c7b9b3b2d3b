"""ORACLE -- test infrastructure only (tests/ may import it; the product never does).

CPU restatement, in plain PyTorch autograd + torch.optim, of the two TRAINING LOOPS of the watermark round trip, so that the
HIP recipe's loss trajectories (tests/roundtrip.py) can be checked step by step for their first steps:

  stage1_steps   train/latent_wm_pretrain.py:164-221  one optimisation step = encoder forward (:172), gen_combined_latents at
                 scale 1.0 (:176-177), frozen VAE decode of raw latents (:100-104,180-181), Identity noiser (:186-189), decoder in
                 train() mode (:160,191), BCE-with-logits against one-hot bits (:194-196), loss = msgloss (:207-213 before epoch
                 7), AdamW(lr, weight_decay 1e-4) on encoder + decoder (:125-128, :220-221)
  ppft_steps     train/ppft_train.py:987-1068  mapper (:990), no-grad encoder residual x 0.18215 (:994-996), add_noise pair
                 (:1010-1011), clean / watermarked U-Net passes (:1026-1035), MSE (:1051), backward (:1058), clip_grad_norm_ of
                 the LoRA parameters at 1.0 (:1059-1065; the mapper is not clipped), AdamW over two groups [LoRA, mapper] with
                 weight decay 1e-2 (:779-787), constant learning rate here

Built from the already pinned / restated pieces: ppft_oracle (U-Net twin of scripts/lib/original_unet.py, LoRA forwards of
utils/lora_modules.py, MapperNet / SecretEncoder of utils/models.py -- PINNED by tests/golden), vae_oracle and decoder_oracle
(diffusers / torchvision architectures restated, UNPINNED: those packages are not on disk).  The optimizers are torch.optim's own,
which is what the reference calls.
"""
import torch
import torch.nn.functional as F

from . import decoder_oracle as D
from . import ppft_oracle as O
from .vae_oracle import VAEOracle

VAE_SCALING = 0.18215


def stage1_steps(vae_sd, vae_cfg, enc_state, dec_state, batches, bits, base_res, res, lr=1e-3, weight_decay=1e-4, grads=None):
    """`enc_state`: {lin_w, lin_b, conv_w, conv_b}; `dec_state`: SecretDecoder.state_dict() (torchvision names under "model.");
    `batches`: [{lat (raw latents), msg, sd_noise [23][B], drop_mask [B,1280]}].  Returns [msgloss per step]; the states are
    updated in place (BatchNorm running statistics included, as F.batch_norm does in training mode).  `grads` (a list) collects the
    encoder's gradients of every step before the optimizer consumes them."""
    enc = {k: v.detach().clone().float().requires_grad_(True) for k, v in enc_state.items()}
    dec = {}
    for k, v in dec_state.items():
        k = k[len("model."):] if k.startswith("model.") else k
        v = v.detach().clone()
        if v.is_floating_point() and "running_" not in k:
            v = v.float().requires_grad_(True)
        dec[k] = v
    params = list(enc.values()) + [v for v in dec.values() if v.requires_grad]
    opt = torch.optim.AdamW(params, lr=lr, weight_decay=weight_decay)
    vae = VAEOracle(vae_sd, vae_cfg)
    losses = []
    for b in batches:
        msg = b["msg"].float()
        wm = O.secret_encoder(msg, enc["lin_w"], enc["lin_b"], enc["conv_w"], enc["conv_b"], base_res, res)
        watermarked = b["lat"] + wm * 1.0                                     # gen_combined_latents without the corner augmentation
        image = vae.decode(watermarked * VAE_SCALING)                         # the oracle's decode divides by the scaling factor again
        logits = D.secret_decoder_train(dec, image, bits, b["sd_noise"], b["drop_mask"])
        labels = F.one_hot(msg.long(), num_classes=2).float()
        msgloss = F.binary_cross_entropy_with_logits(logits, labels)
        opt.zero_grad()
        msgloss.backward()
        if grads is not None:
            grads.append({k: v.grad.detach().clone() for k, v in enc.items()})
        opt.step()
        losses.append(float(msgloss.detach()))
    return losses, {k: v.detach() for k, v in enc.items()}, {k: v.detach() for k, v in dec.items()}


def ppft_steps(unet_sd, unet_cfg, lora, E, enc_state, batches, base_res, res, lr, bf16=True, max_grad_norm=1.0,
               betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8):
    """`lora`: {site: (down, up)}; `E`: MapperNet table [bits, r]; `batches`: [{z, msg, eps, t, ctx}].  bf16=True mirrors the
    bf16 storage of the HIP U-Net (rounding of every activation; the gradients flow through the rounding as through an identity).
    Returns ([loss per step], lora, E) with the trained tensors."""
    lora = {k: (d.detach().clone().float().requires_grad_(True), u.detach().clone().float().requires_grad_(True))
            for k, (d, u) in lora.items()}
    E = E.detach().clone().float().requires_grad_(True)
    lp = [p for pair in lora.values() for p in pair]
    opt = torch.optim.AdamW([{"params": lp, "lr": lr}, {"params": [E], "lr": lr}], betas=betas, weight_decay=weight_decay, eps=eps)
    losses = []
    for b in batches:
        with torch.no_grad():
            wm = O.secret_encoder(b["msg"].float(), enc_state["lin_w"], enc_state["lin_b"], enc_state["conv_w"], enc_state["conv_b"],
                                  base_res, res) * VAE_SCALING
        loss, _, _, _ = O.ppft_loss(unet_sd, unet_cfg, lora, E, b["msg"].float(), b["z"], wm, b["eps"], b["t"], b["ctx"], bf16=bf16)
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(lp, max_grad_norm)
        opt.step()
        losses.append(float(loss.detach()))
    return losses, lora, E
