"""LPIPS(VGG16) on the HIP kernels -- the perceptual loss of stage 1 (reference: ``loss_fn_vgg = lpips.LPIPS(net='vgg')``,
train/latent_wm_pretrain.py:111, and ``loss_fn_vgg(clean_image, watermarked_image).mean()``, :182-190).

The ``lpips`` package (0.1.4) and its weights are third-party and absent from this image; this module follows the published
algorithm (v0.1 linear heads, see csrc/aql_lpips.hip) with the package's state-dict key names, so its checkpoint loads by key:

    net.slice1.{0,2}, net.slice2.{5,7}, net.slice3.{10,12,14}, net.slice4.{17,19,21}, net.slice5.{24,26,28} .weight / .bias
    lin{0..4}.model.1.weight                                   [1, C, 1, 1], C = 64, 128, 256, 512, 512

Activations are bf16 channels-last; the 13 convolutions run on the implicit-GEMM 3x3 kernel (ops.conv3x3: forward and
backward-data), ReLU / max-pool / the per-tap distance are kernels of csrc/aql_lpips.hip.  The two images go through the
network as ONE batch of 2B; the gradient flows to the second image (the watermarked one), like in the reference where the
first is decoded under no_grad.  Parity: oracle/lpips_oracle.py (plain torch fp32 restatement, UNPINNED).
"""
import torch
import torch.nn as nn

from . import _lib as L
from . import ops

VGG16_SLICES = (((0, 3, 64), (2, 64, 64)), ((5, 64, 128), (7, 128, 128)), ((10, 128, 256), (12, 256, 256), (14, 256, 256)),
                ((17, 256, 512), (19, 512, 512), (21, 512, 512)), ((24, 512, 512), (26, 512, 512), (28, 512, 512)))
LPIPS_CHNS = (64, 128, 256, 512, 512)


def lpips_keys():
    """state-dict inventory {key: shape} of lpips.LPIPS(net='vgg') (frozen VGG16 features + the five linear heads)."""
    keys = {}
    for s, convs in enumerate(VGG16_SLICES, start=1):
        for idx, cin, cout in convs:
            keys[f"net.slice{s}.{idx}.weight"] = (cout, cin, 3, 3)
            keys[f"net.slice{s}.{idx}.bias"] = (cout,)
    for i, c in enumerate(LPIPS_CHNS):
        keys[f"lin{i}.model.1.weight"] = (1, c, 1, 1)
    return keys


class _ScaleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.float().contiguous()
        B, C, H, W = x.shape
        y = torch.empty((B, 8, H, W), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
        L.call("aql_lpips_scale", L.ptr(x), B, H, W, L.ptr(y), L.stream_ptr())
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = ops.as_cl(dy)
        B, _, H, W = dy.shape
        dx = torch.empty((B, 3, H, W), dtype=torch.float32, device=dy.device)
        L.call("aql_lpips_scale_bwd", L.ptr(dy), B, H, W, L.ptr(dx), L.stream_ptr())
        return dx


class _ReluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = ops.as_cl(x)
        y = torch.empty_like(x, memory_format=torch.channels_last)
        L.call("aql_relu_bf16", L.ptr(x), x.numel(), L.ptr(y), L.stream_ptr())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = ops.as_cl(dy)
        dx = torch.empty_like(y, memory_format=torch.channels_last)
        L.call("aql_relu_bf16_bwd", L.ptr(dy), L.ptr(y), y.numel(), L.ptr(dx), L.stream_ptr())
        return dx


class _PoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = ops.as_cl(x)
        B, C, H, W = x.shape
        y = torch.empty((B, C, H // 2, W // 2), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        L.call("aql_maxpool2x2_nhwc", L.ptr(x), B, H, W, C, L.ptr(y), L.stream_ptr())
        ctx.save_for_backward(x, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y = ctx.saved_tensors
        dy = ops.as_cl(dy)
        B, C, H, W = x.shape
        dx = torch.empty_like(x, memory_format=torch.channels_last)
        L.call("aql_maxpool2x2_nhwc_bwd", L.ptr(x), L.ptr(y), L.ptr(dy), B, H, W, C, L.ptr(dx), L.stream_ptr())
        return dx


class _TapFn(torch.autograd.Function):
    """One feature tap: f [2B, C, H, W] (first half: reference image) -> per-sample distance contribution [B] (fp32)."""

    @staticmethod
    def forward(ctx, f, w):
        f = ops.as_cl(f)
        B2, C, H, W = f.shape
        B = B2 // 2
        out = torch.zeros(B, dtype=torch.float32, device=f.device)
        L.call("aql_lpips_layer", L.ptr(f[:B]), L.ptr(f[B:]), L.ptr(w), B, H * W, C, L.ptr(out), L.stream_ptr())
        ctx.save_for_backward(f, w)
        return out

    @staticmethod
    def backward(ctx, g):
        f, w = ctx.saved_tensors
        B2, C, H, W = f.shape
        B = B2 // 2
        df = torch.zeros_like(f, memory_format=torch.channels_last)   # no gradient to the reference half
        L.call("aql_lpips_layer_bwd", L.ptr(f[:B]), L.ptr(f[B:]), L.ptr(w), B, H * W, C, L.ptr(g.float().contiguous()),
               L.ptr(df[B:]), L.stream_ptr())
        return df, None


class LPIPS(nn.Module):
    """``LPIPS(state_dict)(img0, img1) -> [B, 1, 1, 1]`` like ``lpips.LPIPS(net='vgg')(in0, in1)`` (inputs in [-1, 1],
    ``normalize=False``; H and W multiples of 32).  Frozen: no parameter receives a gradient; ``img1`` does."""

    def __init__(self, state_dict, device="cuda"):
        super().__init__()
        self.packed = []
        for s, convs in enumerate(VGG16_SLICES, start=1):
            blk = []
            for idx, cin, cout in convs:
                w = state_dict[f"net.slice{s}.{idx}.weight"].to(device).float()
                b = state_dict[f"net.slice{s}.{idx}.bias"].to(device).float()
                blk.append(ops.PackedConv3x3(w, b, 1))
            self.packed.append(blk)
        self.lin = [state_dict[f"lin{i}.model.1.weight"].to(device).float().reshape(-1).contiguous() for i in range(5)]

    def forward(self, in0, in1):
        if not in1.is_cuda:
            raise L.AqlError("LPIPS: the HIP path needs GPU tensors; there is no CPU fallback")
        B, _, H, W = in1.shape
        if H % 32 or W % 32:
            raise ValueError("LPIPS: H and W must be multiples of 32 (four 2x2 max-pools on 16-byte channel vectors)")
        h = _ScaleFn.apply(torch.cat([in0.detach().float(), in1.float()], dim=0))
        total = None
        for s, blk in enumerate(self.packed):
            if s > 0:
                h = _PoolFn.apply(h)
            for pk in blk:
                h = _ReluFn.apply(ops.conv3x3(h, pk))
            d = _TapFn.apply(h, self.lin[s])
            total = d if total is None else total + d
        return total.view(B, 1, 1, 1)


def synthetic_state_dict(seed=2048, device="cpu"):
    """Counter-based stand-in for the VGG16 + LPIPS weights (there is no checkpoint here): He-style conv weights so that
    activations stay O(1) through 13 ReLU layers, non-negative linear heads like the trained ones."""
    from . import synth
    sd = {}
    for k, shp in lpips_keys().items():
        if k.startswith("lin"):
            sd[k] = synth.normal("lpips." + k, shp, 1.0, seed, device).abs() / shp[1]
        elif k.endswith("bias"):
            sd[k] = synth.normal("lpips." + k, shp, 0.05, seed, device)
        else:
            sd[k] = synth.normal("lpips." + k, shp, (2.0 / (shp[1] * 9)) ** 0.5, seed, device)
    return sd
