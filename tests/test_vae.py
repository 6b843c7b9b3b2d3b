"""Frozen SD-1.5 VAE.  CPU: key inventory of the state dict and self-consistency of the
oracle (tests/ may import oracle/).  GPU: the HIP encode / decode against the oracle on a reduced-width VAE and on the
full-size encoder at 256x256 (tolerances below; the oracle is unpinned -- diffusers is not on disk)."""
import pytest
import torch

from aqualora_amd import synth
from aqualora_amd.vae import SD15_VAE, encode_gflop, synthetic_state_dict, vae_keys
from oracle.vae_oracle import VAEOracle

TINY_VAE = dict(SD15_VAE, block_out_channels=(32, 64, 64, 64))


def test_vae_key_inventory_matches_sd15():
    keys = vae_keys()
    # diffusers AutoencoderKL (SD-1.5): 248 tensors, 83.65 M parameters
    assert len(keys) == 248
    n = sum(torch.Size(s).numel() for s in keys.values())
    assert abs(n - 83_653_863) == 0, n
    assert keys["encoder.down_blocks.1.resnets.0.conv_shortcut.weight"] == (256, 128, 1, 1)
    assert keys["encoder.mid_block.attentions.0.to_q.weight"] == (512, 512)
    assert keys["decoder.up_blocks.3.resnets.2.conv2.weight"] == (128, 128, 3, 3)
    assert keys["quant_conv.weight"] == (8, 8, 1, 1) and keys["post_quant_conv.weight"] == (4, 4, 1, 1)
    assert "encoder.down_blocks.3.downsamplers.0.conv.weight" not in keys
    assert 1100 < encode_gflop() < 1150      # ~1.13 TFLOP per 512x512 image


def test_vae_oracle_shapes_and_posterior():
    sd = synthetic_state_dict(TINY_VAE)
    o = VAEOracle(sd, TINY_VAE)
    x = synth.normal("vae.x", (2, 3, 32, 32), 0.5, 7).clamp(-1, 1)
    mean, logvar = o.encode_moments(x)
    assert mean.shape == (2, 4, 4, 4) and logvar.shape == (2, 4, 4, 4)
    noise = synth.normal("vae.n", (2, 4, 4, 4), 1.0, 7)
    z = o.encode(x, noise)
    assert torch.allclose(z, (mean + torch.exp(0.5 * logvar) * noise) * 0.18215)
    assert torch.equal(o.encode(x, sample=False), mean * 0.18215)
    img = o.decode(z)
    assert img.shape == (2, 3, 32, 32) and torch.isfinite(img).all()
    # the asymmetric pad of the encoder's Downsample2D: a stride-1 padding-1 convolution sampled at odd positions
    w, b = sd["encoder.down_blocks.0.downsamplers.0.conv.weight"], sd["encoder.down_blocks.0.downsamplers.0.conv.bias"]
    h = synth.normal("vae.h", (1, 32, 8, 8), 1.0, 7)
    a = torch.nn.functional.conv2d(torch.nn.functional.pad(h, (0, 1, 0, 1)), w, b, stride=2)
    c = torch.nn.functional.conv2d(h, w, b, padding=1)[:, :, 1::2, 1::2]
    assert torch.allclose(a, c, atol=1e-5)


def _rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max() / b.abs().max())


@pytest.mark.gpu
def test_vae_hip_vs_oracle_tiny():
    """Reduced-width VAE, 64x64 image: HIP vs the bf16-mirroring oracle (tight) and vs the fp32 oracle (bf16 storage)."""
    from aqualora_amd.vae import AutoencoderKL
    sd = synthetic_state_dict(TINY_VAE)
    vae = AutoencoderKL(sd, TINY_VAE, "cuda")
    x = synth.normal("vae.x", (2, 3, 64, 64), 0.5, 7).clamp(-1, 1)
    mean, logvar = vae.encode_moments(x.cuda())
    om, ol = VAEOracle(sd, TINY_VAE, bf16=True).encode_moments(x)
    fm, fl = VAEOracle(sd, TINY_VAE).encode_moments(x)
    assert mean.shape == (2, 4, 8, 8)
    assert _rel(mean, om) < 2e-2 and _rel(logvar, ol) < 2e-2, (_rel(mean, om), _rel(logvar, ol))
    assert _rel(mean, fm) < 5e-2 and _rel(logvar, fl) < 5e-2
    noise = synth.normal("vae.n", (2, 4, 8, 8), 1.0, 7)
    z = vae.encode(x.cuda(), noise.cuda())
    assert _rel(z, (fm + torch.exp(0.5 * fl) * noise) * 0.18215) < 5e-2
    img = vae.decode(z)
    oi = VAEOracle(sd, TINY_VAE, bf16=True).decode(z.cpu())
    assert img.shape == (2, 3, 64, 64)
    assert _rel(img, oi) < 3e-2, _rel(img, oi)
    # determinism
    assert torch.equal(vae.encode_moments(x.cuda())[0], mean)


@pytest.mark.gpu
def test_vae_full_size_encoder_vs_oracle():
    """Full-width SD-1.5 encoder at 256x256 (attention over 1024 tokens, all four resolutions, both 1x1 shortcuts)."""
    from aqualora_amd.vae import AutoencoderKL
    sd = synthetic_state_dict(SD15_VAE)
    vae = AutoencoderKL(sd, SD15_VAE, "cuda")
    x = synth.normal("vae.x", (1, 3, 256, 256), 0.5, 11).clamp(-1, 1)
    mean, logvar = vae.encode_moments(x.cuda())
    om, ol = VAEOracle(sd, SD15_VAE, bf16=True).encode_moments(x)
    assert mean.shape == (1, 4, 32, 32)
    assert _rel(mean, om) < 3e-2 and _rel(logvar, ol) < 3e-2, (_rel(mean, om), _rel(logvar, ol))


@pytest.mark.gpu
def test_vae_decode_is_differentiable_wrt_latents():
    """Stage 1 back-propagates through the frozen decoder (latent_wm_pretrain.py:180-181): d<image, w>/dz from the HIP
    graph (conv backward-data incl. upsample adjoint, GroupNorm backward, GEMM-built wide-head attention backward)
    against torch autograd on the fp32 oracle."""
    from aqualora_amd.vae import AutoencoderKL
    sd = synthetic_state_dict(TINY_VAE)
    vae = AutoencoderKL(sd, TINY_VAE, "cuda")
    z0 = synth.normal("vae.z", (2, 4, 8, 8), 0.18215, 5)
    w = synth.normal("vae.w", (2, 3, 64, 64), 1.0, 5)
    z = z0.clone().cuda().requires_grad_(True)
    img = vae.decode_grad(z)
    (img * w.cuda()).sum().backward()
    zr = z0.clone().requires_grad_(True)
    ref = VAEOracle(sd, TINY_VAE).decode(zr)
    (ref * w).sum().backward()
    assert _rel(img, ref) < 5e-2
    g, gr = z.grad.float().cpu(), zr.grad
    l2 = float((g - gr).norm() / gr.norm())
    assert l2 < 6e-2, l2          # bf16 storage of every activation and gradient along ~40 ops
    assert z.grad.shape == z0.shape and torch.isfinite(z.grad).all()


@pytest.mark.gpu
def test_stage1_step_through_the_hip_vae_decoder():
    """latent_wm_pretrain.py:164-221 with ``decode_latents = vae.decode`` on the HIP VAE (raw latents, :100-104): the step
    runs end to end on the GPU and the SecretEncoder receives finite, non-zero gradients through the frozen decoder."""
    from aqualora_amd import noise as NZ, stage1 as S1
    from aqualora_amd.decoder import SecretDecoder
    from aqualora_amd.vae import AutoencoderKL
    from aqualora_amd.watermark import SecretEncoder
    vae = AutoencoderKL(synthetic_state_dict(TINY_VAE), TINY_VAE, "cuda")
    enc = SecretEncoder(48, base_res=4, resolution=8).cuda()
    with torch.no_grad():
        enc.secret_scaler[5].weight.copy_(synth.normal("s1.conv", (4, 4, 3, 3), 0.05, 3))
    dec = SecretDecoder(48).cuda().train()
    step = S1.Stage1Step(enc, dec, lambda z: vae.decode_grad(z, scaled=False), NZ.Noiser(["Identity"], [1.0]))
    lat = synth.normal("s1.lat", (2, 4, 8, 8), 1.0, 3).cuda()
    msg = synth.bits("s1.msg", (2, 48), 3).cuda()
    out = step.losses(lat, msg, noiser_choice=[1.0])
    loss = out["loss"] if isinstance(out, dict) else out[0]
    loss.backward()
    g = enc.secret_scaler[0].weight.grad
    assert g is not None and torch.isfinite(g).all() and float(g.abs().max()) > 0
    # the full stage-1 loss of the late phase (lpips * 5 + msgloss + prvl * 1.5, latent_wm_pretrain.py:207-209) with the HIP
    # LPIPS(VGG16): the perceptual term reaches the SecretEncoder through the frozen VAE decoder as well
    from aqualora_amd.lpips import LPIPS, synthetic_state_dict as lpips_sd
    enc.zero_grad()
    step2 = S1.Stage1Step(enc, dec, lambda z: vae.decode_grad(z, scaled=False), NZ.Noiser(["Identity"], [1.0]),
                          lpips_fn=LPIPS(lpips_sd(), "cuda"))
    step2.warmup = False
    out2 = step2.losses(lat, msg, epochs_done=11, noiser_choice=[1.0])
    assert float(out2["lpips_loss"]) > 0
    (out2["lpips_loss"] * 5).backward()
    g2 = enc.secret_scaler[0].weight.grad
    assert g2 is not None and torch.isfinite(g2).all() and float(g2.abs().max()) > 0
