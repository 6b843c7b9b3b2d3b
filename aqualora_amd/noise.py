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
