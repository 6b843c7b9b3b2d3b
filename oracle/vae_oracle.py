"""CPU oracle for the frozen SD-1.5 VAE (TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; the product path aqualora_amd/vae.py never touches it).

Plain-PyTorch restatement of diffusers 0.24 ``AutoencoderKL`` (reference call sites train/ppft_train.py:538-541,993,
train/latent_wm_pretrain.py:171,180-181, evaluation/utils_eval.py:74-106).  **Parity UNPINNED**: diffusers is not under
/root/reference and the reference holds no vectors for the VAE; the architecture below is its published one:

  Encoder : conv_in 3->128 | 4 x DownEncoderBlock2D (128,256,512,512; 2 ResnetBlock2D each, no time embedding;
            Downsample2D = F.pad(x,(0,1,0,1)) + Conv2d(3, stride 2, padding 0) on all but the last) |
            UNetMidBlock2D (Resnet, Attention(1 head of 512, GroupNorm first, residual), Resnet) |
            GroupNorm(32, 1e-6) + SiLU + conv_out 512->8 ; then quant_conv 1x1 8->8
  Posterior: mean, logvar = chunk(moments, 2, 1); logvar clamped to [-30, 20]; sample = mean + exp(logvar/2) * eps
  Decoder : post_quant_conv 1x1 4->4 | conv_in 4->512 | UNetMidBlock2D | 4 x UpDecoderBlock2D (512,512,256,128; 3 Resnets
            each; Upsample2D = nearest x2 + Conv2d(3, padding 1) on all but the last) | GroupNorm + SiLU + conv_out 128->3
  ResnetBlock2D: h = conv1(silu(norm1(x))); h = conv2(silu(norm2(h))); out = shortcut(x) + h   (1x1 conv_shortcut iff
            channels change; output_scale_factor 1, dropout 0)

``bf16=True`` mirrors the HIP path's storage: weights and every op output are rounded to bf16 (accumulation in fp32).
"""
import torch
import torch.nn.functional as F


def _rb(t, on):
    return t.to(torch.bfloat16).float() if on else t


class VAEOracle:
    def __init__(self, sd, cfg, bf16=False):
        self.cfg, self.bf16 = cfg, bf16
        self.sd = {k: _rb(v.detach().float().cpu(), bf16) for k, v in sd.items()}

    def r(self, t):
        return _rb(t, self.bf16)

    def conv(self, x, key, stride=1, padding=1):
        w = self.sd[key + ".weight"]
        if w.dim() == 2:
            w = w[:, :, None, None]
        return self.r(F.conv2d(x, w, self.sd[key + ".bias"], stride=stride, padding=padding))

    def norm(self, x, key, silu):
        y = F.group_norm(x, self.cfg["norm_groups"], self.sd[key + ".weight"], self.sd[key + ".bias"], self.cfg["eps"])
        return self.r(F.silu(y) if silu else y)

    def resnet(self, x, p):
        h = self.conv(self.norm(x, p + ".norm1", True), p + ".conv1")
        h = self.norm(h, p + ".norm2", True)
        sc = x
        if (p + ".conv_shortcut.weight") in self.sd:
            sc = self.conv(x, p + ".conv_shortcut", padding=0)
        w = self.sd[p + ".conv2.weight"]
        return self.r(F.conv2d(h, w, self.sd[p + ".conv2.bias"], padding=1) + sc)

    def lin(self, t, key):
        w = self.sd[key + ".weight"]
        return self.r(t @ w.reshape(w.shape[0], -1).t() + self.sd[key + ".bias"])

    def attn(self, x, p):
        B, C, H, W = x.shape
        t = self.norm(x, p + ".group_norm", False).permute(0, 2, 3, 1).reshape(B, H * W, C)
        q, k, v = (self.lin(t, f"{p}.{n}") for n in ("to_q", "to_k", "to_v"))
        prob = self.r(torch.softmax((q @ k.transpose(1, 2)) * C ** -0.5, dim=-1))
        o = self.r(prob @ v)
        w = self.sd[p + ".to_out.0.weight"]
        y = self.r(o @ w.reshape(C, -1).t() + self.sd[p + ".to_out.0.bias"] + x.permute(0, 2, 3, 1).reshape(B, H * W, C))
        return y.reshape(B, H, W, C).permute(0, 3, 1, 2)

    def mid(self, x, p):
        x = self.resnet(x, p + ".resnets.0")
        x = self.attn(x, p + ".attentions.0")
        return self.resnet(x, p + ".resnets.1")

    def encode_moments(self, x):
        ch, L_ = self.cfg["block_out_channels"], self.cfg["layers_per_block"]
        h = self.conv(self.r(x.float()), "encoder.conv_in")
        for i in range(len(ch)):
            for j in range(L_):
                h = self.resnet(h, f"encoder.down_blocks.{i}.resnets.{j}")
            if i + 1 < len(ch):
                h = self.conv(F.pad(h, (0, 1, 0, 1)), f"encoder.down_blocks.{i}.downsamplers.0.conv", stride=2, padding=0)
        h = self.mid(h, "encoder.mid_block")
        h = self.conv(self.norm(h, "encoder.conv_norm_out", True), "encoder.conv_out")
        m = self.conv(h, "quant_conv", padding=0)
        mean, logvar = m.chunk(2, dim=1)
        return mean, logvar.clamp(-30.0, 20.0)

    def encode(self, x, noise=None, sample=True):
        mean, logvar = self.encode_moments(x)
        z = mean + torch.exp(0.5 * logvar) * noise if sample else mean
        return z * self.cfg["scaling_factor"]

    def decode(self, z_scaled):
        rch, L_ = tuple(reversed(self.cfg["block_out_channels"])), self.cfg["layers_per_block"]
        z = self.r(z_scaled.float() / self.cfg["scaling_factor"])
        h = self.conv(z, "post_quant_conv", padding=0)
        h = self.conv(h, "decoder.conv_in")
        h = self.mid(h, "decoder.mid_block")
        for i in range(len(rch)):
            for j in range(L_ + 1):
                h = self.resnet(h, f"decoder.up_blocks.{i}.resnets.{j}")
            if i + 1 < len(rch):
                h = self.conv(F.interpolate(h, scale_factor=2.0, mode="nearest"), f"decoder.up_blocks.{i}.upsamplers.0.conv")
        return self.conv(self.norm(h, "decoder.conv_norm_out", True), "decoder.conv_out")
