"""Shared builders for the parity tests: synthetic tiny U-Net state, LoRA weights and PPFT inputs by name."""
import torch

from aqualora_amd import synth
from aqualora_amd.unet import UNet2DConditionModel, init_synthetic, lora_keys

SEED = 2048
TINY = dict(block_out_channels=(32, 64, 64, 64), cross_attention_dim=32, attention_heads=2, layers_per_block=2)
TINY_RANK = 8
LORA_CASES = [("lin_a", 320, 320, 16, 8), ("lin_b", 768, 320, 77, 8), ("lin_c", 320, 2560, 8, 32),
              ("lin_d", 1280, 320, 8, 32)]


def T(name, shape, std=1.0, device="cpu"):
    return synth.normal(name, shape, std, SEED, device)


def tiny_unet(device="cpu", dtype=torch.float32, cfg=None):
    unet = UNet2DConditionModel(cfg or TINY, device=device, dtype=dtype)
    init_synthetic(unet, SEED)
    return unet


def tiny_lora(keys, unet, rank=TINY_RANK, up_std=0.1):
    """{key: (down_w, up_w)} with the same names/shapes as tests/golden/make_golden.py."""
    out = {}
    for k in keys:
        m = unet.get_submodule(k)
        if hasattr(m, "in_channels"):
            ds, us = (rank, m.in_channels, 1, 1), (m.out_channels, rank, 1, 1)
        else:
            ds, us = (rank, m.in_features), (m.out_features, rank)
        out[k] = (synth.normal(k + ".lora.down", ds, 1.0 / rank, SEED), synth.normal(k + ".lora.up", us, up_std, SEED))
    return out


def ppft_inputs(cfg=TINY, B=2, bits=48, res=16, rank=TINY_RANK, device="cpu"):
    return dict(E=T("ppft.mapper.E", (bits, rank), device=device), msg=synth.bits("ppft.msg", (B, bits), SEED, device),
                z=T("ppft.z", (B, 4, res, res), device=device), wm=T("ppft.wm", (B, 4, res, res), 0.5, device),
                eps=T("ppft.eps", (B, 4, res, res), device=device), t=synth.randint("ppft.t", (B,), 1000, SEED, device),
                ctx=T("ppft.ctx", (B, 77, cfg["cross_attention_dim"]), device=device))


def prvl_case(i, B, H, W, amp):
    """inputs of the PRVL golden cases (same construction as tests/golden/make_golden.py:prvl_case)."""
    a = T(f"prvl.a{i}", (B, 3, H, W), 0.5)
    d = T(f"prvl.d{i}", (B, 3, H, W), amp)
    d[:, :, H // 3: H // 3 + 20, W // 4: W // 4 + 25] *= 4.0
    return a, (a + d)
