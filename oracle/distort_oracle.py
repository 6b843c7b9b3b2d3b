"""CPU restatement (torch, fp32) of the kornia-based distortions the reference calls -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module; the product path
(aqualora_amd/noise.py) never does and fails loudly without the HIP library.

PARITY UNPINNED: kornia 0.6.12 (requirements.txt:13) is a third-party dependency that is neither under /root/reference
nor installed in this image, and the reference holds no tests or golden vectors for these layers.  The functions below
restate kornia 0.6.12's published algorithms tensor-op by tensor-op:
  * ``color_jiggle``  kornia.augmentation.ColorJiggle.apply_transform -> enhance.adjust_{brightness,contrast,
    saturation,hue} + color.rgb_to_hsv / hsv_to_rgb   (call sites: noises.py:97-103, noiser.py:52-57,
    utils_eval.py:271-276)
  * ``rotate``        kornia.geometry.transform.rotate -> affine -> warp_affine (bilinear, zeros, align_corners=True)
    (call sites: noises.py:29, utils_eval.py:292)
  * ``sharpness``     kornia.enhance.sharpness (call sites: noises.py:117, utils_eval.py:294)
They are written with differentiable torch ops so that autograd supplies the reference gradients for the adjoint tests.
"""
import math

import torch
import torch.nn.functional as F


def rgb_to_hsv(image, eps=1e-8):
    max_rgb, argmax_rgb = image.max(-3)
    min_rgb = image.min(-3)[0]
    deltac = max_rgb - min_rgb
    v = max_rgb
    s = deltac / (max_rgb + eps)
    deltac = torch.where(deltac == 0, torch.ones_like(deltac), deltac)
    rc, gc, bc = torch.unbind(max_rgb.unsqueeze(-3) - image, dim=-3)
    h1 = bc - gc
    h2 = (rc - bc) + 2.0 * deltac
    h3 = (gc - rc) + 4.0 * deltac
    h = torch.stack((h1, h2, h3), dim=-3) / deltac.unsqueeze(-3)
    h = torch.gather(h, dim=-3, index=argmax_rgb.unsqueeze(-3)).squeeze(-3)
    h = (h / 6.0) % 1.0
    h = 2.0 * math.pi * h
    return torch.stack((h, s, v), dim=-3)


def hsv_to_rgb(image):
    h = image[..., 0, :, :] / (2 * math.pi)
    s = image[..., 1, :, :]
    v = image[..., 2, :, :]
    hi = torch.floor(h * 6) % 6
    f = ((h * 6) % 6) - hi
    one = torch.tensor(1.0)
    p = v * (one - s)
    q = v * (one - f * s)
    t = v * (one - (one - f) * s)
    hi = hi.long()
    indices = torch.stack([hi, hi + 6, hi + 12], dim=-3)
    out = torch.stack((v, q, p, p, t, v, t, v, v, q, p, p, p, p, t, v, v, q), dim=-3)
    return torch.gather(out, -3, indices)


def _per_sample(f, x):
    return torch.as_tensor(f, dtype=x.dtype).reshape(-1).expand(x.shape[0]).reshape(-1, 1, 1, 1)


def adjust_brightness(x, factor):  # additive, clamped
    return (x + _per_sample(factor, x)).clamp(0.0, 1.0)


def adjust_contrast(x, factor):  # multiplicative, clamped (ColorJiggle's legacy contrast)
    return (x * _per_sample(factor, x)).clamp(0.0, 1.0)


def adjust_saturation(x, factor):
    hsv = rgb_to_hsv(x)
    h, s, v = torch.chunk(hsv, 3, dim=-3)
    s = (s * _per_sample(factor, x)).clamp(0.0, 1.0)
    return hsv_to_rgb(torch.cat([h, s, v], dim=-3))


def adjust_hue(x, factor_rad):
    hsv = rgb_to_hsv(x)
    h, s, v = torch.chunk(hsv, 3, dim=-3)
    h = torch.fmod(h + _per_sample(factor_rad, x), 2 * math.pi)
    return hsv_to_rgb(torch.cat([h, s, v], dim=-3))


def color_jiggle(x, brightness, contrast, saturation, hue, order=(0, 1, 2, 3)):
    """x [B,3,H,W] in [0,1]; factors as ColorJiggle samples them; hue in turns (multiplied by 2*pi like kornia does)."""
    ops = [lambda t: adjust_brightness(t, torch.as_tensor(brightness, dtype=t.dtype) - 1.0),
           lambda t: adjust_contrast(t, contrast),
           lambda t: adjust_saturation(t, saturation),
           lambda t: adjust_hue(t, torch.as_tensor(hue, dtype=t.dtype) * 2 * math.pi)]
    for idx in order:
        x = ops[int(idx)](x)
    return x


def rotate(x, angle_deg):
    """Anti-clockwise rotation about ((W-1)/2, (H-1)/2): OpenCV-style matrix, inverse-mapped sampling grid."""
    B, C, H, W = x.shape
    a = torch.as_tensor(angle_deg, dtype=x.dtype).reshape(-1).expand(B) * (math.pi / 180.0)
    cs, sn = torch.cos(a), torch.sin(a)
    cx, cy = (W - 1) / 2.0, (H - 1) / 2.0
    ys, xs = torch.meshgrid(torch.arange(H, dtype=x.dtype), torch.arange(W, dtype=x.dtype), indexing="ij")
    dx, dy = xs[None] - cx, ys[None] - cy
    sx = cs[:, None, None] * dx - sn[:, None, None] * dy + cx
    sy = sn[:, None, None] * dx + cs[:, None, None] * dy + cy
    grid = torch.stack([sx / (W - 1) * 2 - 1, sy / (H - 1) * 2 - 1], dim=-1)
    return F.grid_sample(x, grid, mode="bilinear", padding_mode="zeros", align_corners=True)


def sharpness(x, factor):
    B, C, H, W = x.shape
    f = torch.as_tensor(factor, dtype=x.dtype).reshape(-1).expand(B)
    kernel = torch.tensor([[1.0, 1, 1], [1, 5, 1], [1, 1, 1]], dtype=x.dtype).view(1, 1, 3, 3).repeat(C, 1, 1, 1) / 13
    degenerate = F.conv2d(x, kernel, bias=None, stride=1, groups=C).clamp(0.0, 1.0)
    mask = torch.ones_like(degenerate)
    padded_mask = F.pad(mask, [1, 1, 1, 1])
    padded_degenerate = F.pad(degenerate, [1, 1, 1, 1])
    result = torch.where(padded_mask == 1, padded_degenerate, x)
    outs = []
    for i in range(B):  # _blend_one per sample
        fi = float(f[i])
        if fi == 0.0:
            outs.append(result[i])
        elif fi == 1.0:
            outs.append(x[i])
        else:
            res = result[i] + (x[i] - result[i]) * fi
            outs.append(res if 0.0 < fi < 1.0 else res.clamp(0.0, 1.0))
    return torch.stack(outs)
