"""Distortion layers of the watermark pipeline (reference utils/noise_layers/*), HIP-backed.

``Identity`` (identity.py:3-11) and ``JpegCompression`` (jpeg_compression.py:67-162) have their arithmetic in the
reference tree itself and are pinned by golden vectors (DCT-mask JPEG simulation, differentiable, forward AND backward
through one kernel because the layer is a fixed linear map per 8x8x3 block).  The kornia / torchvision based layers
(noises.py: CropandResize, GaussianBlur, GaussianNoise, ColorJitter, Rotation, Sharpness; noiser.py:46-71 and
utils_eval.py:269-301 ``distorsion_unit``) follow the published kornia 0.6.12 / torchvision 0.15.2 behaviour -- those
packages are not in this image, so their parity is UNPINNED (checked against oracle/distort_oracle.py only).
Calling convention follows the reference: ``layer([image, cover]) -> [image', cover]`` with NCHW fp32.  Random parameters
are drawn on the host where the reference draws them; every deterministic image map and its adjoint is a HIP kernel.
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
    """Normalised 1-D Gaussian window (kornia ``get_gaussian_kernel1d``).  ``sigma``: a float -> [ksize]; a [B] tensor
    (one sigma per sample, as kornia's RandomGaussianBlur draws them) -> [B, ksize]."""
    x = torch.arange(ksize, dtype=torch.float32) - (ksize - 1) / 2
    if torch.is_tensor(sigma):
        sg = sigma.detach().float().cpu().reshape(-1, 1)
        g = torch.exp(-(x[None] * x[None]) / (2.0 * sg * sg))
        return (g / g.sum(dim=1, keepdim=True)).to(device).contiguous()
    g = torch.exp(-(x * x) / (2.0 * sigma * sigma))
    return (g / g.sum()).to(device)


class _BlurFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, taps_x, taps_y):
        x = x.float().contiguous()
        B, C, H, W = x.shape
        y, tmp = torch.empty_like(x), torch.empty_like(x)
        per = int(taps_x.dim() == 2)
        L.call("aql_gauss_blur2", L.ptr(x), L.ptr(y), L.ptr(tmp), B, C, H, W, taps_x.shape[-1], taps_y.shape[-1], L.ptr(taps_x),
               L.ptr(taps_y), per, 0, L.stream_ptr())
        ctx.save_for_backward(taps_x, taps_y)
        return y

    @staticmethod
    def backward(ctx, dy):
        taps_x, taps_y = ctx.saved_tensors
        dy = dy.float().contiguous()
        B, C, H, W = dy.shape
        dx, tmp = torch.empty_like(dy), torch.empty_like(dy)
        per = int(taps_x.dim() == 2)
        L.call("aql_gauss_blur2", L.ptr(dy), L.ptr(dx), L.ptr(tmp), B, C, H, W, taps_x.shape[-1], taps_y.shape[-1],
               L.ptr(taps_x), L.ptr(taps_y), per, 1, L.stream_ptr())
        return dx, None, None


def gaussian_blur(x, ksize, sigma):
    """kornia ``gaussian_blur2d(x, (ky, kx), (sigma, sigma), border_type='reflect')``.  ``ksize``: int (square) or (ky, kx)
    like kornia's kernel_size; ``sigma``: float, or a [B] tensor with one value per sample."""
    if not x.is_cuda:
        raise L.AqlError("gaussian_blur: the HIP path needs a GPU tensor; there is no CPU fallback")
    ky, kx = (int(ksize), int(ksize)) if isinstance(ksize, int) or not hasattr(ksize, "__len__") else map(int, ksize)
    if torch.is_tensor(sigma):
        if sigma.numel() != x.shape[0]:
            raise ValueError(f"per-sample sigma needs {x.shape[0]} values, got {sigma.numel()}")
    else:
        sigma = float(sigma)
    return _BlurFn.apply(x, gaussian_taps(kx, sigma, x.device), gaussian_taps(ky, sigma, x.device))


class _GaussNoiseFn(torch.autograd.Function):
    """y = x + std * noise [clamped to [0,1]]: kornia's RandomGaussianNoise is differentiable in x (stage 1 back-propagates
    msgloss through it to the SecretEncoder, latent_wm_pretrain.py:186-196); the clamp passes gradient inside (0,1)."""

    @staticmethod
    def forward(ctx, x, noise, std, clamp01):
        x = x.float().contiguous()
        y = torch.empty_like(x)
        L.call("aql_add_gauss_noise", L.ptr(x), L.ptr(noise), float(std), int(clamp01), x.numel(), L.ptr(y), L.stream_ptr())
        ctx.clamp01 = bool(clamp01)
        if clamp01:
            ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        if not ctx.clamp01:
            return dy, None, None, None
        (y,) = ctx.saved_tensors
        return dy * ((y > 0) & (y < 1)).to(dy.dtype), None, None, None


def add_gaussian_noise(x, std, clamp01=False, noise=None):
    if not x.is_cuda:
        raise L.AqlError("add_gaussian_noise: the HIP path needs a GPU tensor; there is no CPU fallback")
    noise = torch.randn(x.shape, device=x.device, dtype=torch.float32) if noise is None else noise.float().contiguous()
    return _GaussNoiseFn.apply(x, noise, float(std), bool(clamp01))


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
    """noises.py:59-70: kornia 0.6.12 ``RandomGaussianBlur((3, 9), (0, max), p=1)`` -- a FIXED anisotropic kernel of 3 rows x
    9 columns (kornia's kernel_size is (ky, kx)), reflect border, one sigma ~ U(0, max) per SAMPLE (recalled from the
    published implementation; kornia is not on disk)."""

    def __init__(self, blur=2.0):
        super().__init__()
        self.gaussian_blur_max = blur

    def forward(self, noised_and_cover):
        x = noised_and_cover[0]
        sigma = torch.from_numpy(np.random.rand(x.shape[0]) * self.gaussian_blur_max).float().clamp_min(1e-3)
        noised_and_cover[0] = gaussian_blur(x, (3, 9), sigma)
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


def _need_gpu(x, what):
    if not x.is_cuda:
        raise L.AqlError(f"{what}: the HIP path needs a GPU tensor; there is no CPU fallback")


class _ColorJiggleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, factors, order):
        x = x.float().contiguous()
        B, C, H, W = x.shape
        assert C == 3, "colour jiggle works on RGB images"
        y = torch.empty_like(x)
        L.call("aql_color_jiggle", L.ptr(x), None, L.ptr(y), B, H, W, L.ptr(factors), L.ptr(order), L.stream_ptr())
        ctx.save_for_backward(x, factors, order)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, factors, order = ctx.saved_tensors
        dy = dy.float().contiguous()
        B, C, H, W = x.shape
        dx = torch.empty_like(x)
        L.call("aql_color_jiggle", L.ptr(x), L.ptr(dy), L.ptr(dx), B, H, W, L.ptr(factors), L.ptr(order), L.stream_ptr())
        return dx, None, None


def color_jiggle(x, brightness, contrast, saturation, hue, order=(0, 1, 2, 3)):
    """kornia ColorJiggle with explicit parameters: x [B,3,H,W] in [0,1]; brightness/contrast/saturation/hue are [B]
    factors as kornia samples them (brightness & co. around 1, hue in [-0.5, 0.5] turns); ``order`` is the permutation
    of (0 brightness, 1 contrast, 2 saturation, 3 hue) kornia draws with randperm."""
    _need_gpu(x, "color_jiggle")
    f = torch.stack([torch.as_tensor(brightness, dtype=torch.float32) - 1.0, torch.as_tensor(contrast, dtype=torch.float32),
                     torch.as_tensor(saturation, dtype=torch.float32),
                     torch.as_tensor(hue, dtype=torch.float32) * (2.0 * np.pi)], dim=-1).reshape(-1, 4)
    f = f.expand(x.shape[0], 4).contiguous().to(x.device)
    o = torch.as_tensor(list(order), dtype=torch.int32, device=x.device)
    assert sorted(o.tolist()) == [0, 1, 2, 3]
    return _ColorJiggleFn.apply(x, f, o)


def random_color_jiggle(x, brightness, contrast, saturation, hue):
    """ColorJiggle(brightness=(lo,hi), ..., p=1): per-sample uniform factors, one random op order per call."""
    B = x.shape[0]
    u = lambda r: torch.empty(B).uniform_(r[0], r[1])
    return color_jiggle(x, u(brightness), u(contrast), u(saturation), u(hue), torch.randperm(4).tolist())


class _RotateFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, angles):
        x = x.float().contiguous()
        B, C, H, W = x.shape
        y = torch.empty_like(x)
        L.call("aql_rotate_bilinear", L.ptr(x), L.ptr(y), B, C, H, W, L.ptr(angles), 0, L.stream_ptr())
        ctx.save_for_backward(angles)
        return y

    @staticmethod
    def backward(ctx, dy):
        (angles,) = ctx.saved_tensors
        dy = dy.float().contiguous()
        B, C, H, W = dy.shape
        dx = torch.empty_like(dy)
        L.call("aql_rotate_bilinear", L.ptr(dy), L.ptr(dx), B, C, H, W, L.ptr(angles), 1, L.stream_ptr())
        return dx, None


def rotate(x, angle_deg):
    """kornia rotate(): anti-clockwise by angle_deg ([B] or scalar) about the centre, bilinear, zero padding."""
    _need_gpu(x, "rotate")
    a = torch.as_tensor(angle_deg, dtype=torch.float32).reshape(-1).expand(x.shape[0]).contiguous().to(x.device)
    return _RotateFn.apply(x, a)


class _SharpnessFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, factor):
        x = x.float().contiguous()
        B, C, H, W = x.shape
        y = torch.empty_like(x)
        L.call("aql_sharpness", L.ptr(x), None, L.ptr(y), None, B, C, H, W, L.ptr(factor), L.stream_ptr())
        ctx.save_for_backward(x, factor)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, factor = ctx.saved_tensors
        dy = dy.float().contiguous()
        B, C, H, W = x.shape
        dx, tmp = torch.empty_like(x), torch.empty_like(x)
        L.call("aql_sharpness", L.ptr(x), L.ptr(dy), L.ptr(dx), L.ptr(tmp), B, C, H, W, L.ptr(factor), L.stream_ptr())
        return dx, None


def sharpness(x, factor):
    """kornia sharpness(): x [B,C,H,W] in [0,1], factor [B] or scalar."""
    _need_gpu(x, "sharpness")
    f = torch.as_tensor(factor, dtype=torch.float32).reshape(-1).expand(x.shape[0]).contiguous().to(x.device)
    return _SharpnessFn.apply(x, f)


class ColorJitter(nn.Module):
    """noises.py:88-104: images in [-1,1] -> [0,1] -> ColorJiggle(b .7-1.3, c .8-1.25, s .8-1.25, h +-.2) -> [-1,1]."""

    def forward(self, noised_and_cover):
        y = random_color_jiggle(noised_and_cover[0] / 2 + 0.5, (0.7, 1.3), (0.8, 1.25), (0.8, 1.25), (-0.2, 0.2))
        noised_and_cover[0] = y * 2 - 1
        return noised_and_cover


class Rotation(nn.Module):
    """noises.py:20-31: RandomRotation(degrees) -> angle ~ U(-degrees, degrees) per sample."""

    def __init__(self, degrees=180):
        super().__init__()
        self.degrees = degrees

    def forward(self, noised_and_cover):
        x = noised_and_cover[0]
        noised_and_cover[0] = rotate(x, torch.empty(x.shape[0]).uniform_(-self.degrees, self.degrees))
        return noised_and_cover


class Sharpness(nn.Module):
    """noises.py:106-119: strength ~ U(0, max) on the host, then RandomSharpness(strength) -> factor ~ U(0, strength)."""

    def __init__(self, strength=1.0):
        super().__init__()
        self.strength_max = strength

    def forward(self, noised_and_cover):
        x = noised_and_cover[0]
        strength = np.random.rand() * self.strength_max
        y = sharpness(x / 2 + 0.5, torch.empty(x.shape[0]).uniform_(0, max(strength, 1e-12)))
        noised_and_cover[0] = y * 2 - 1
        return noised_and_cover


class Noiser(nn.Module):
    """noiser.py:12-44: one layer per call, chosen with the given probabilities; layer names as in the reference."""

    def __init__(self, noise_layers, posibilities, device=None):
        super().__init__()
        self.noise_layers = [Identity()]
        for layer in noise_layers:
            if isinstance(layer, str):
                if layer == "Identity":
                    continue
                elif layer == "Jpeg":
                    self.noise_layers.append(JpegCompression(device))
                elif layer == "CropandResize":
                    self.noise_layers.append(CropandResize((256, 512), (256, 512)))
                elif layer == "GaussianBlur":
                    self.noise_layers.append(GaussianBlur(10.0))
                elif layer == "GaussianNoise":
                    self.noise_layers.append(GaussianNoise(0.2))
                elif layer == "ColorJitter":
                    self.noise_layers.append(ColorJitter())
                else:
                    raise ValueError("Wrong layer placeholder string in Noiser.__init__().")
            else:
                self.noise_layers.append(layer)
        self.posibilities = posibilities

    def forward(self, encoded_and_cover, possibilites=None):
        p = self.posibilities if possibilites is None else possibilites
        idx = int(np.random.choice(len(self.noise_layers), 1, p=p)[0])
        return self.noise_layers[idx](encoded_and_cover)


class RobNoiser:
    """rob_enhance_finetune.py:121-132: Identity / color_jitter / crop / blur / noise on [0,1] images."""

    distorsion_types = ["Identity", "color_jitter", "crop", "blur", "noise"]

    def __init__(self, posibilities):
        self.posibilities = posibilities

    def __call__(self, encoded_image, possibilites=None):
        p = self.posibilities if possibilites is None else possibilites
        kind = self.distorsion_types[int(np.random.choice(len(self.distorsion_types), 1, p=p)[0])]
        return encoded_image if kind == "Identity" else distorsion_unit(encoded_image, kind)


def eval_distorsion_unit(encoded_image, type):
    """evaluation/utils_eval.py:269-301, the tensor-space attacks: 'color_jitter' (.9-1.1, hue +-.1), 'crop' (460x460
    window -> back to the input size), 'blur' (k=3, sigma 4), 'noise' (std .1, clamp), 'rotation' (15 deg), 'sharpness'
    (factor ~ U(0,10)).  'jpeg_compress' (PIL codec) and 'SDEdit' (a diffusion pipeline) are outside this library."""
    x = encoded_image if encoded_image.dim() == 4 else encoded_image[None]
    if type == "color_jitter":
        y = random_color_jiggle(x, (0.9, 1.1), (0.9, 1.1), (0.9, 1.1), (-0.1, 0.1))
    elif type == "crop":
        H, W = x.shape[2:]
        top, left = np.random.randint(0, H - 460 + 1), np.random.randint(0, W - 460 + 1)
        y = crop_resize(x, top, left, 460, 460, H, W)
    elif type == "blur":
        y = gaussian_blur(x, 3, 4.0)
    elif type == "noise":
        y = add_gaussian_noise(x, 0.1, clamp01=True)
    elif type == "rotation":
        y = rotate(x, 15.0)
    elif type == "sharpness":
        y = sharpness(x, torch.empty(x.shape[0]).uniform_(0, 10.0))
    else:
        raise ValueError("Wrong distorsion type in add_distorsion().")
    return y if encoded_image.dim() == 4 else y[0]


def distorsion_unit(encoded_image, type):
    """noiser.py:46-71 (rob-finetune): 'color_jitter' (.8-1.2, hue +-.1), 'crop' (432..512 window -> 512x512), 'blur'
    (3 x 5 kernel, sigma 4), 'noise' (std 0.1, clamp to [0,1])."""
    if type == "crop":
        ch, cw = np.random.randint(432, 512), np.random.randint(432, 512)
        H, W = encoded_image.shape[2:]
        top, left = np.random.randint(0, H - ch + 1), np.random.randint(0, W - cw + 1)
        return crop_resize(encoded_image, top, left, ch, cw, 512, 512)
    if type == "blur":   # RandomGaussianBlur((3, 5), (4.0, 4.0)): fixed 3-row x 5-column kernel, sigma 4
        return gaussian_blur(encoded_image, (3, 5), 4.0)
    if type == "noise":
        return add_gaussian_noise(encoded_image, 0.1, clamp01=True)
    if type == "color_jitter":
        return random_color_jiggle(encoded_image, (0.8, 1.2), (0.8, 1.2), (0.8, 1.2), (-0.1, 0.1))
    raise ValueError("Wrong distorsion type.")
