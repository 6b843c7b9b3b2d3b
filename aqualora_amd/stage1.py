"""Stage 1 -- latent watermark pre-training (reference train/latent_wm_pretrain.py:159-225) and the robustness fine-tune
step of the message decoder (train/rob_enhance_finetune.py:996-1036), HIP-backed.

What is built here: ``PRVL_loss`` (:42-50), ``gen_combined_latents`` (:133-149), the step's loss schedule (:196-217), the
trainable SecretEncoder / SecretDecoder (forward AND backward in HIP, see watermark.py / decoder.py) and the distortion
layers (noise.py).  ``decode_latents`` is a differentiable callable: ``lambda z: vae.decode_grad(z, scaled=False)`` of
``aqualora_amd.vae.AutoencoderKL`` runs the frozen decoder and its backward on the HIP kernels (the reference decodes raw,
unscaled latents at :100-104).  What stays outside (third-party and absent from this image): LPIPS (``lpips_fn`` callable,
optional) and the diffusion sampling pipeline that feeds rob-finetune with images.
"""
import random

import torch

from . import _lib as L
from .decoder import bce_with_logits
from .noise import crop_resize

WINDOW_SIZE = 32  # latent_wm_pretrain.py:38


class _PrvlFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img1, img2, win):
        a, b = img1.float().contiguous(), img2.float().contiguous()
        B, C, H, W = a.shape
        Wo = W + 2 * (win // 2) - win + 1
        scratch = torch.empty(B * H * W + B * H * Wo + 3 * 1024 + 2, device=a.device, dtype=torch.float32)
        loss = torch.empty((), device=a.device, dtype=torch.float32)
        arg = torch.empty((), device=a.device, dtype=torch.int64)
        L.call("aql_prvl_loss_fwd", L.ptr(a), L.ptr(b), B, C, H, W, win, L.ptr(scratch), L.ptr(loss), L.ptr(arg),
               L.stream_ptr())
        ctx.save_for_backward(a, b, arg)
        ctx.win = win
        return loss

    @staticmethod
    def backward(ctx, g):
        a, b, arg = ctx.saved_tensors
        B, C, H, W = a.shape
        d1 = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        d2 = torch.empty_like(a) if ctx.needs_input_grad[1] else None
        L.call("aql_prvl_loss_bwd", L.ptr(a), L.ptr(b), L.ptr(arg), L.ptr(g.float().contiguous()), B, C, H, W, ctx.win,
               L.ptr(d1), L.ptr(d2), L.stream_ptr())
        return d1, d2, None


def PRVL_loss(img1, img2):
    """latent_wm_pretrain.py:42-50: max over every 32x32 window (padding 16, all samples) of mean(|img1 - img2|)."""
    if not img1.is_cuda:
        raise L.AqlError("PRVL_loss: the HIP path needs GPU tensors; there is no CPU fallback")
    return _PrvlFn.apply(img1, img2, WINDOW_SIZE)


def gen_combined_latents(latents, wm_latent, scale=1.0, cornerfy_aug=None, scales=None):
    """latent_wm_pretrain.py:133-149.  With probability 1/4 the watermark's four corner quadrants are pasted into a
    zero canvas that is 1-2x larger and the canvas is bilinearly shrunk back (the watermark survives a centre stretch);
    otherwise latents + wm*scale.  ``cornerfy_aug`` / ``scales`` pin the random draws (python ``random``, like the
    reference)."""
    if cornerfy_aug is None:
        cornerfy_aug = random.choice([True, False, False, False])
    height, width = wm_latent.shape[2], wm_latent.shape[3]
    if cornerfy_aug:
        hs, ws = scales if scales is not None else (random.uniform(1.0, 2.0), random.uniform(1.0, 2.0))
        Hc, Wc = int(height * hs), int(width * ws)      # F.interpolate(scale_factor) output size = floor(in * scale)
        canvas = torch.zeros(wm_latent.shape[0], wm_latent.shape[1], Hc, Wc, device=wm_latent.device,
                             dtype=wm_latent.dtype)
        h2, w2 = height // 2, width // 2
        canvas[:, :, :h2, :w2] = wm_latent[:, :, :h2, :w2]
        canvas[:, :, :h2, -w2:] = wm_latent[:, :, :h2, -w2:]
        canvas[:, :, -h2:, :w2] = wm_latent[:, :, -h2:, :w2]
        canvas[:, :, -h2:, -w2:] = wm_latent[:, :, -h2:, -w2:]
        wm_template = crop_resize(canvas, 0, 0, Hc, Wc, height, width)
    else:
        wm_template = wm_latent
    return latents + wm_template * scale


class Stage1Step:
    """One optimisation step body of latent_wm_pretrain.py:164-221 (everything between ``zero_grad`` and ``backward``).

    ``decode_latents(latents) -> image in [-1,1]`` and ``lpips_fn(img1, img2) -> [B,...]`` are the third-party pieces
    (VAE decoder, LPIPS) and are supplied by the caller; ``noiser`` is ``noise.Noiser``.
    """

    def __init__(self, sec_encoder, sec_decoder, decode_latents, noiser, lpips_fn=None):
        self.sec_encoder, self.sec_decoder = sec_encoder, sec_decoder
        self.decode_latents, self.noiser, self.lpips_fn = decode_latents, noiser, lpips_fn
        self.msgloss_10buffer = []
        self.warmup = True

    def new_epoch(self):
        """Call at the start of every epoch: the reference re-creates the 10-batch message-loss window per epoch
        (latent_wm_pretrain.py:163), so the warm-up exit test never mixes batches of two epochs."""
        self.msgloss_10buffer = []

    def losses(self, latents, msg, epochs_done=0, resumed=False, combine=None, noiser_choice=None):
        _, wm_latent = self.sec_encoder(latents, msg.float())
        combine = dict(combine or {})
        watermarked_latents = gen_combined_latents(latents, wm_latent, scale=0.03 if self.warmup else 1.0, **combine)
        with torch.no_grad():
            clean_image = self.decode_latents(latents)
        watermarked_image = self.decode_latents(watermarked_latents)
        zero = torch.zeros((), device=latents.device)
        lpips_loss = self.lpips_fn(clean_image, watermarked_image).mean() if self.lpips_fn is not None else zero
        prvl_loss = PRVL_loss(clean_image, watermarked_image)
        if noiser_choice is not None:
            probs = noiser_choice
        elif epochs_done > 12 or resumed:
            probs = [0.4, 0.1, 0.2, 0.05, 0.1, 0.15]
        else:
            probs = [0.6, 0.0, 0.4, 0.0, 0.0, 0.0]
        distorted = self.noiser([watermarked_image, None], probs)[0]
        reveal_output = self.sec_decoder(distorted)
        labels = torch.nn.functional.one_hot(msg.long(), num_classes=2).float()
        msgloss = bce_with_logits(reveal_output, labels)
        # :199-206 -- ten consecutive batches under 0.1 end the warm-up
        self.msgloss_10buffer = (self.msgloss_10buffer + [float(msgloss.detach())])[-10:]
        if len(self.msgloss_10buffer) == 10 and sum(self.msgloss_10buffer) / 10 < 0.1:
            self.warmup = False
        if self.warmup:
            loss = msgloss
        elif epochs_done > 10 or resumed:
            loss = lpips_loss * 5 + msgloss * 1.0 + prvl_loss * 1.5
        elif epochs_done > 6:
            loss = lpips_loss + msgloss
        else:
            loss = msgloss
        return {"loss": loss, "msgloss": msgloss, "lpips_loss": lpips_loss, "prvl_loss": prvl_loss,
                "logits": reveal_output, "watermarked_image": watermarked_image}


def prepare_rob_finetune(msgdecoder, process_group=None, n_buckets=4):
    """What ``accelerator.prepare(msgdecoder)`` does in rob_enhance_finetune.py:917-919 besides moving the module: DDP's
    construction-time broadcast of rank 0's parameters and buffers, so that every rank fine-tunes the same decoder, and DDP's
    grad-ready buckets: under torch.distributed the decoder gets a `dp.ModuleGradExchange` (flat gradient buffer, bucket
    all-reduces launched from gradient hooks under the rest of backward -- through aql_comm_* when the trainer's RCCL
    communicator is in use, AQL_COMM=1, else torch.distributed's asynchronous all-reduce), which `rob_finetune_step` drives."""
    from . import dp
    dp.broadcast_module_(msgdecoder, process_group)
    if dp.exchange_active(process_group):
        comm, _ = dp.make_comm(process_group)
        msgdecoder._aql_exchange = dp.ModuleGradExchange(msgdecoder, process_group, comm, n_buckets)
    return msgdecoder


def rob_finetune_step(msgdecoder, optimizer, images01, secret_bits, distort=None, process_group=None):
    """rob_enhance_finetune.py:1018-1036: generated images in [0,1] (no grad) -> distortion -> [-1,1] -> msgdecoder ->
    BCE against one-hot bits -> backward -> optimizer step.  Returns (loss, bit accuracy).  Under torch.distributed (one
    process per GPU, the reference wraps the decoder in DDP through accelerate) rank 0's BatchNorm buffers are broadcast
    before the forward (one collective per dtype) and the gradients are mean-all-reduced over RCCL: in buckets launched from
    gradient hooks DURING backward when `prepare_rob_finetune` installed the exchange, else in flat buckets after it."""
    from . import dp
    dp.broadcast_buffers_(msgdecoder, process_group)
    ex = getattr(msgdecoder, "_aql_exchange", None)
    x = images01.detach().float()
    if distort is not None:
        x = distort(x)
    x = (x * 2 - 1).detach()
    logits = msgdecoder(x)
    decoded = torch.argmax(logits, dim=-1)
    acc = (decoded == secret_bits.long()).float().mean()
    loss = bce_with_logits(logits.float(), torch.nn.functional.one_hot(secret_bits.long(), num_classes=2).float())
    loss.backward()
    if ex is not None:
        ex.finish()
        optimizer.step()
        ex.zero_grad()
    else:
        dp.allreduce_module_grads_(msgdecoder.parameters(), process_group)
        optimizer.step()
        optimizer.zero_grad()
    return loss.detach(), acc
