"""Distortion layers of the watermark pipeline (reference utils/noise_layers/*), HIP-backed.

Built so far: ``Identity`` (identity.py:3-11) and ``JpegCompression`` (jpeg_compression.py:67-162) -- the layer whose
arithmetic lives in the reference tree itself (DCT-mask JPEG simulation, differentiable, forward AND backward through
one kernel because the layer is a fixed linear map per 8x8x3 block).  The kornia / torchvision based layers
(crop-resize, blur, noise, colour jitter) are not built yet (their arithmetic is third-party and
unpinned here).  Calling convention follows the reference: ``layer([image, cover]) -> [image', cover]`` with NCHW fp32.
"""
import torch
import torch.nn as nn

from . import _lib as L


class Identity(nn.Module):
    def forward(self, noised_and_cover):
        return noised_and_cover


class _JpegFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, keep):
        if not x.is_cuda:
            raise L.AqlError("JpegCompression: the HIP path needs a GPU tensor; there is no CPU fallback")
        x = x.float().contiguous()
        B, C, H, W = x.shape
        assert C == 3, "JPEG simulation works on RGB images"
        y = torch.empty_like(x)
        L.call("aql_jpeg_mask", L.ptr(x), L.ptr(y), B, H, W, keep[0], keep[1], keep[2], 0, L.stream_ptr())
        ctx.keep = keep
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.float().contiguous()
        B, C, H, W = dy.shape
        dx = torch.empty_like(dy)
        k = ctx.keep
        L.call("aql_jpeg_mask", L.ptr(dy), L.ptr(dx), B, H, W, k[0], k[1], k[2], 1, L.stream_ptr())
        return dx, None


class JpegCompression(nn.Module):
    def __init__(self, device=None, yuv_keep_weights=(25, 9, 9)):
        super().__init__()
        self.yuv_keep_weighs = tuple(int(k) for k in yuv_keep_weights)

    def forward(self, noised_and_cover):
        noised_and_cover[0] = _JpegFn.apply(noised_and_cover[0], self.yuv_keep_weighs)
        return noised_and_cover


# ------------------------------------------------------------------------------------------------------------
# Deterministic image maps behind CropandResize / GaussianBlur / GaussianNoise (noises.py:34-85) and the rob-finetune
# distorsion_unit (noiser.py:46-71).  Random parameters are drawn with numpy exactly where the reference draws them.
import numpy as np  # noqa: E402


class _CropResizeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, top, left, ch, cw, oh, ow):
        x = x.float().contiguous()
        B, C, H, W = x.shape
        y = torch.empty(B, C, oh, ow, dtype=torch.float32, device=x.device)
        L.call("aql_crop_resize_bilinear", L.ptr(x), L.ptr(y), B * C, H, W, top, left, ch, cw, oh, ow, 0, L.stream_ptr())
        ctx.cfg = (B, C, H, W, top, left, ch, cw, oh, ow)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, C, H, W, top, left, ch, cw, oh, ow = ctx.cfg
        dy = dy.float().contiguous()
        dx = torch.empty(B, C, H, W, dtype=torch.float32, device=dy.device)
        L.call("aql_crop_resize_bilinear", L.ptr(dy), L.ptr(dx), B * C, H, W, top, left, ch, cw, oh, ow, 1, L.stream_ptr())
        return dx, None, None, None, None, None, None


def crop_resize(x, top, left, ch, cw, oh, ow):
    if not x.is_cuda:
        raise L.AqlError("crop_resize: the HIP path needs a GPU tensor; there is no CPU fallback")
    return _CropResizeFn.apply(x, int(top), int(left), int(ch), int(cw), int(oh), int(ow))


def gaussian_taps(ksize, sigma, device):
    x = torch.arange(ksize, dtype=torch.float32) - (ksize - 1) / 2
    g = torch.exp(-(x * x) / (2.0 * sigma * sigma))
    return (g / g.sum()).to(device)


class _BlurFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, taps):
        x = x.float().contiguous()
        B, C, H, W = x.shape
        y, tmp = torch.empty_like(x), torch.empty_like(x)
        L.call("aql_gauss_blur", L.ptr(x), L.ptr(y), L.ptr(tmp), B * C, H, W, taps.numel(), L.ptr(taps), 0, L.stream_ptr())
        ctx.save_for_backward(taps)
        return y

    @staticmethod
    def backward(ctx, dy):
        (taps,) = ctx.saved_tensors
        dy = dy.float().contiguous()
        B, C, H, W = dy.shape
        dx, tmp = torch.empty_like(dy), torch.empty_like(dy)
        L.call("aql_gauss_blur", L.ptr(dy), L.ptr(dx), L.ptr(tmp), B * C, H, W, taps.numel(), L.ptr(taps), 1, L.stream_ptr())
        return dx, None


def gaussian_blur(x, ksize, sigma):
    if not x.is_cuda:
        raise L.AqlError("gaussian_blur: the HIP path needs a GPU tensor; there is no CPU fallback")
    return _BlurFn.apply(x, gaussian_taps(int(ksize), float(sigma), x.device))


def add_gaussian_noise(x, std, clamp01=False, noise=None):
    if not x.is_cuda:
        raise L.AqlError("add_gaussian_noise: the HIP path needs a GPU tensor; there is no CPU fallback")
    x = x.float().contiguous()
    noise = torch.randn_like(x) if noise is None else noise.float().contiguous()
    y = torch.empty_like(x)
    L.call("aql_add_gauss_noise", L.ptr(x), L.ptr(noise), float(std), int(clamp01), x.numel(), L.ptr(y), L.stream_ptr())
    return y


class CropandResize(nn.Module):
    """noises.py:34-57: random crop, resize to a random size, resize back to 512x512 (all bilinear, no antialias)."""

    def __init__(self, crop_size_range, resize_size_range, out_size=512):
        super().__init__()
        self.cmin, self.cmax = crop_size_range
        self.rmin, self.rmax = resize_size_range
        self.out_size = out_size

    def forward(self, noised_and_cover):
        x = noised_and_cover[0]
        ch, cw = np.random.randint(self.cmin, self.cmax), np.random.randint(self.cmin, self.cmax)
        rh, rw = np.random.randint(self.rmin, self.rmax), np.random.randint(self.rmin, self.rmax)
        H, W = x.shape[2:]
        top, left = np.random.randint(0, H - ch + 1), np.random.randint(0, W - cw + 1)
        y = crop_resize(x, top, left, ch, cw, rh, rw)
        noised_and_cover[0] = crop_resize(y, 0, 0, rh, rw, self.out_size, self.out_size)
        return noised_and_cover


class GaussianBlur(nn.Module):
    """noises.py:59-70 (kornia RandomGaussianBlur((3,9),(0,max)) -- kernel size and sigma sampling recalled)."""

    def __init__(self, blur=2.0):
        super().__init__()
        self.gaussian_blur_max = blur

    def forward(self, noised_and_cover):
        k = int(np.random.choice([3, 5, 7, 9]))
        sigma = max(1e-3, np.random.rand() * self.gaussian_blur_max)
        noised_and_cover[0] = gaussian_blur(noised_and_cover[0], k, sigma)
        return noised_and_cover


class GaussianNoise(nn.Module):
    """noises.py:72-85: x + N(0, std), std ~ U(0, max)."""

    def __init__(self, std=0.1):
        super().__init__()
        self.gaussian_std_max = std

    def forward(self, noised_and_cover):
        std = np.random.rand() * self.gaussian_std_max
        noised_and_cover[0] = add_gaussian_noise(noised_and_cover[0], std)
        return noised_and_cover


def distorsion_unit(encoded_image, type):
    """noiser.py:46-71 (rob-finetune): 'crop' (432..512 window -> 512x512), 'blur' (k in 3..5, sigma 4), 'noise'
    (std 0.1, clamp to [0,1]).  'color_jitter' (kornia ColorJiggle) is not built yet."""
    if type == "crop":
        ch, cw = np.random.randint(432, 512), np.random.randint(432, 512)
        H, W = encoded_image.shape[2:]
        top, left = np.random.randint(0, H - ch + 1), np.random.randint(0, W - cw + 1)
        return crop_resize(encoded_image, top, left, ch, cw, 512, 512)
    if type == "blur":
        return gaussian_blur(encoded_image, int(np.random.choice([3, 5])), 4.0)
    if type == "noise":
        return add_gaussian_noise(encoded_image, 0.1, clamp01=True)
    if type == "color_jitter":
        raise NotImplementedError("kornia ColorJiggle is not built yet")
    raise ValueError("Wrong distorsion type.")
