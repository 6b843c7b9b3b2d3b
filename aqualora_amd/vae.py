"""Frozen SD-1.5 VAE (diffusers ``AutoencoderKL``) on the HIP kernels: encode (train/ppft_train.py:993,
``vae.encode(pixel_values).latent_dist.sample() * 0.18215``) and decode (evaluation/utils_eval.py pipelines,
train/latent_wm_pretrain.py:171,180-181 forward), SURVEY.md §8 row A17 / (f) rank 1.

The VAE is frozen everywhere in the reference: weights never get a gradient.  ``encode`` / ``decode`` run without a graph;
``decode_grad`` keeps the autograd graph back to the latents, because stage 1 trains the SecretEncoder THROUGH the decoder
(latent_wm_pretrain.py:180-181) -- every backward op is a HIP kernel too.  It reuses the U-Net's kernels -- implicit-GEMM 3x3
convolutions (with the nearest-x2 upsample folded into the gather on the decoder side and the encoder's
``F.pad(x, (0,1,0,1))`` + stride-2 convolution as a leading-pad-0 variant of the same loader), GroupNorm(32, 1e-6)+SiLU,
bf16 GEMMs with bias/residual epilogues -- plus a row-softmax kernel for the mid-block's single-head 512-wide
attention, which runs as  S = Q K^T (fp32) -> softmax -> P V  on the GEMM kernels (the flash kernels stop at d = 160).

State-dict keys follow diffusers 0.24 (``encoder.down_blocks.0.resnets.0.norm1.weight`` ...,
``encoder.mid_block.attentions.0.to_q.weight``; the pre-0.15 names ``query/key/value/proj_attn`` are accepted too).
diffusers is not on disk: the architecture is restated from its published definition and checked against
``oracle/vae_oracle.py`` (parity UNPINNED, like every third-party component).
"""
import math

import torch

from . import _lib as L
from . import ops, synth

SD15_VAE = dict(in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512),
                layers_per_block=2, norm_groups=32, eps=1e-6, scaling_factor=0.18215)
CL = torch.channels_last
_OLD_ATTN = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}


def vae_keys(cfg=SD15_VAE):
    """{key: shape} of the diffusers AutoencoderKL state dict for ``cfg`` (encoder, decoder, quant convs)."""
    ch, L_ = cfg["block_out_channels"], cfg["layers_per_block"]
    zc = cfg["latent_channels"]
    out = {}

    def conv(p, co, ci, k):
        out[p + ".weight"], out[p + ".bias"] = (co, ci, k, k), (co,)

    def norm(p, c):
        out[p + ".weight"], out[p + ".bias"] = (c,), (c,)

    def resnet(p, ci, co):
        norm(p + ".norm1", ci)
        conv(p + ".conv1", co, ci, 3)
        norm(p + ".norm2", co)
        conv(p + ".conv2", co, co, 3)
        if ci != co:
            conv(p + ".conv_shortcut", co, ci, 1)

    def mid(p, c):
        resnet(p + ".resnets.0", c, c)
        norm(p + ".attentions.0.group_norm", c)
        for n in ("to_q", "to_k", "to_v", "to_out.0"):
            out[f"{p}.attentions.0.{n}.weight"], out[f"{p}.attentions.0.{n}.bias"] = (c, c), (c,)
        resnet(p + ".resnets.1", c, c)

    conv("encoder.conv_in", ch[0], cfg["in_channels"], 3)
    ci = ch[0]
    for i, co in enumerate(ch):
        for j in range(L_):
            resnet(f"encoder.down_blocks.{i}.resnets.{j}", ci, co)
            ci = co
        if i + 1 < len(ch):
            conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", co, co, 3)
    mid("encoder.mid_block", ch[-1])
    norm("encoder.conv_norm_out", ch[-1])
    conv("encoder.conv_out", 2 * zc, ch[-1], 3)
    conv("quant_conv", 2 * zc, 2 * zc, 1)
    conv("post_quant_conv", zc, zc, 1)
    rch = tuple(reversed(ch))
    conv("decoder.conv_in", rch[0], zc, 3)
    mid("decoder.mid_block", rch[0])
    ci = rch[0]
    for i, co in enumerate(rch):
        for j in range(L_ + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}", ci, co)
            ci = co
        if i + 1 < len(rch):
            conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", co, co, 3)
    norm("decoder.conv_norm_out", rch[-1])
    conv("decoder.conv_out", cfg["out_channels"], rch[-1], 3)
    return out


def synthetic_state_dict(cfg=SD15_VAE, seed=2048, device="cpu"):
    """Counter-based synthetic weights (std 1/sqrt(fan_in), GroupNorm gamma 1 / beta 0): there is no checkpoint here."""
    sd = {}
    for k, shp in vae_keys(cfg).items():
        if ".norm" in k or "group_norm" in k or "conv_norm_out" in k:
            sd[k] = torch.ones(shp, device=device) if k.endswith("weight") else torch.zeros(shp, device=device)
        elif k.endswith("bias"):
            sd[k] = synth.normal("vae." + k, shp, 0.02, seed, device)
        else:
            fan_in = math.prod(shp[1:])
            sd[k] = synth.normal("vae." + k, shp, fan_in ** -0.5, seed, device)
    return sd


def _tn_f32(U, V):
    """U[M,P]^T V[M,Q] in fp32 on the transpose-read GEMM (both operands token-major)."""
    C = torch.zeros(U.shape[1], V.shape[1], dtype=torch.float32, device=U.device)
    L.call("aql_gemm_tn_tr_f32", L.ptr(U), U.stride(0), L.ptr(V), V.stride(0), U.shape[0], U.shape[1], V.shape[1], 1.0,
           L.ptr(C), C.stride(0), L.stream_ptr())
    return C


class WideHeadAttentionFn(torch.autograd.Function):
    """softmax(Q K^T / sqrt(C)) V with ONE head of width C (512 in the VAE) on token-major q/k/v [B*N, C], built from
    GEMMs because the flash kernels stop at d = 160:
      forward : S = Q K^T (fp32, aql_gemm_nt_f32_accum) -> P = softmax(S / sqrt(C)) (aql_softmax_rows) -> O = P V
      backward: dV = P^T dO, dK = dS^T Q (aql_gemm_tn_tr_f32);  dP = dO V^T;  dS = P * (dP - rowsum(P dP)) / sqrt(C)
                (aql_softmax_rows_bwd);  dQ = dS K
    P [N,N] bf16 per sample is saved for backward (33 MB at N = 4096)."""

    @staticmethod
    def forward(ctx, q, k, v, B):
        N, C = q.shape[0] // B, q.shape[1]
        dev, st = q.device, L.stream_ptr()
        ws = ops.workspace(dev, max(64 << 20, N * N))   # fp32 scores of one sample (768x768 images: 9216 tokens, 340 MB)
        o = torch.empty_like(q)
        vt = torch.empty(C, N, dtype=torch.bfloat16, device=dev)
        probs = torch.empty(B, N, N, dtype=torch.bfloat16, device=dev)
        for b in range(B):
            sl = slice(b * N, (b + 1) * N)
            s = torch.zeros(N, N, dtype=torch.float32, device=dev)
            L.call("aql_gemm_nt_f32_accum", L.ptr(q[sl]), C, L.ptr(k[sl]), C, N, N, C, 1.0, L.ptr(s), N, L.ptr(ws),
                   ws.numel() * 4, st)
            L.call("aql_softmax_rows", L.ptr(s), N, N, N, float(C ** -0.5), L.ptr(probs[b]), N, st)
            L.call("aql_transpose_bf16", L.ptr(v[sl]), N, C, C, L.ptr(vt), st)
            ops.gemm_bf16(probs[b], vt, out=o[sl])
        ctx.save_for_backward(q, k, v, probs)
        ctx.B = B
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, probs = ctx.saved_tensors
        B = ctx.B
        N, C = q.shape[0] // B, q.shape[1]
        dev, st = q.device, L.stream_ptr()
        do = do.contiguous()
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        kt = torch.empty(C, N, dtype=torch.bfloat16, device=dev)
        ds = torch.empty(N, N, dtype=torch.bfloat16, device=dev)
        for b in range(B):
            sl = slice(b * N, (b + 1) * N)
            dv[sl] = _tn_f32(probs[b], do[sl]).to(torch.bfloat16)
            dp = ops.gemm_bf16(do[sl], v[sl])                       # dO V^T  [N, N]
            L.call("aql_softmax_rows_bwd", L.ptr(probs[b]), L.ptr(dp), N, N, N, float(C ** -0.5), L.ptr(ds), st)
            L.call("aql_transpose_bf16", L.ptr(k[sl]), N, C, C, L.ptr(kt), st)
            ops.gemm_bf16(ds, kt, out=dq[sl])                       # dS K
            dk[sl] = _tn_f32(ds, q[sl]).to(torch.bfloat16)          # dS^T Q
        return dq, dk, dv, None


class AutoencoderKL:
    """``encode(x).latent_dist``-style moments and ``decode(z)`` of the SD-1.5 VAE; all activations bf16 channels-last."""

    def __init__(self, state_dict, cfg=SD15_VAE, device="cuda"):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise L.AqlError("AutoencoderKL needs an MI355X (cuda device); there is no CPU path")
        self.cfg, self.device = dict(cfg), dev
        sd = {}
        for k, v in state_dict.items():
            for old, new in _OLD_ATTN.items():
                k = k.replace(f".attentions.0.{old}.", f".attentions.0.{new}.")
            sd[k] = v
        missing = [k for k in vae_keys(cfg) if k not in sd]
        if missing:
            raise L.AqlError(f"AutoencoderKL: state dict lacks {len(missing)} keys, e.g. {missing[:3]}")
        self.p = {}
        for k, shp in vae_keys(cfg).items():
            if not k.endswith(".weight"):
                continue
            base = k[:-7]
            w, b = sd[k].to(dev).float(), sd[base + ".bias"].to(dev).float()
            if tuple(w.shape) != tuple(shp) and not (len(shp) == 2 and w.dim() == 4):
                raise L.AqlError(f"AutoencoderKL: {k} has shape {tuple(w.shape)}, expected {shp}")
            if len(shp) == 1:
                self.p[base] = (w.to(torch.bfloat16).contiguous(), b.to(torch.bfloat16).contiguous())
            elif len(shp) == 4 and shp[-1] == 3:
                self.p[base] = ops.PackedConv3x3(w, b, 2 if "downsamplers" in base else 1)
            else:   # 1x1 conv or attention linear (old checkpoints store those as 1x1 convs too)
                self.p[base] = ops.PackedLinear(w.reshape(w.shape[0], -1), b)

    # ------------------------------------------------------------------------------------------------ ops
    def _conv3(self, x, key, upsample=False, residual=None, pad_lo=1):
        pk = self.p[key]
        if pad_lo == 1:   # the U-Net's differentiable op (backward-data incl. the folded upsample's adjoint)
            return ops.conv3x3(x, pk, upsample, None, residual)
        B, C, H, W = x.shape
        if C != pk.Cin:   # conv_in: zero-pad 3 (or 4) channels to 8
            xp = x.new_zeros((B, pk.Cin, H, W)).contiguous(memory_format=CL)
            xp[:, :C] = x
            x = xp
        Hl, Wl = (2 * H, 2 * W) if upsample else (H, W)
        Ho, Wo = (Hl + 2 - 3) // pk.stride + 1, (Wl + 2 - 3) // pk.stride + 1
        y = torch.empty((B, pk.Cout, Ho, Wo), dtype=torch.bfloat16, device=x.device, memory_format=CL)
        ws = ops.workspace(x.device)
        for b0, nb in ops.span_chunks(B, max(H * W * pk.Cin, Ho * Wo * pk.Cout) * 2):   # 1 GiB buffer descriptors: sample chunks
            L.call("aql_conv3x3_fwd_pad", L.ptr(x[b0:b0 + nb]), nb, H, W, pk.Cin, L.ptr(pk.wk), L.ptr(pk.bias), pk.Cout, pk.stride,
                   int(upsample), pad_lo, None, 0, L.ptr(None if residual is None else residual[b0:b0 + nb]), L.ptr(y[b0:b0 + nb]),
                   L.ptr(ws), ws.numel() * 4, L.stream_ptr())
        return y[:, :pk.Cout_real] if pk.Cout_real != pk.Cout else y

    def _norm(self, x, key, silu):
        g, b = self.p[key]
        return ops.groupnorm_silu(x, g, b, self.cfg["eps"], silu)

    @staticmethod
    def _tokens(x):
        B, C, H, W = x.shape
        return x.permute(0, 2, 3, 1).reshape(B * H * W, C)   # a view: the map is channels-last

    def _lin(self, t, key, residual=None):
        return ops.lora_linear(t, self.p[key], residual=residual)

    def _resnet(self, x, p):
        h = self._conv3(self._norm(x, p + ".norm1", True), p + ".conv1")
        h = self._norm(h, p + ".norm2", True)
        if (p + ".conv_shortcut") in self.p:
            B, _, H, W = x.shape
            sc = self._lin(self._tokens(x), p + ".conv_shortcut").view(B, H, W, -1).permute(0, 3, 1, 2)
        else:
            sc = x
        return self._conv3(h, p + ".conv2", residual=sc)

    def _attn(self, x, p):
        """diffusers Attention with one head of width C, GroupNorm first, residual last."""
        B, C, H, W = x.shape
        t = self._tokens(self._norm(x, p + ".group_norm", False))
        q, k, v = (self._lin(t, f"{p}.{n}") for n in ("to_q", "to_k", "to_v"))
        o = WideHeadAttentionFn.apply(q, k, v, B)
        y = self._lin(o, p + ".to_out.0", residual=self._tokens(x))
        return y.view(B, H, W, C).permute(0, 3, 1, 2)

    def _mid(self, x, p):
        x = self._resnet(x, p + ".resnets.0")
        x = self._attn(x, p + ".attentions.0")
        return self._resnet(x, p + ".resnets.1")

    # ------------------------------------------------------------------------------------------------ API
    @torch.no_grad()
    def encode_moments(self, x):
        """x [B,3,H,W] in [-1,1] -> (mean, logvar) fp32 [B,4,H/8,W/8] each (``vae.encode(x).latent_dist`` parameters)."""
        ch, L_ = self.cfg["block_out_channels"], self.cfg["layers_per_block"]
        h = x.to(self.device, torch.bfloat16).contiguous(memory_format=CL)
        h = self._conv3(h, "encoder.conv_in")
        for i in range(len(ch)):
            for j in range(L_):
                h = self._resnet(h, f"encoder.down_blocks.{i}.resnets.{j}")
            if i + 1 < len(ch):
                h = self._conv3(h, f"encoder.down_blocks.{i}.downsamplers.0.conv", pad_lo=0)
        h = self._mid(h, "encoder.mid_block")
        h = self._conv3(self._norm(h, "encoder.conv_norm_out", True), "encoder.conv_out")
        B, C2, Hh, Wh = h.shape
        m = self._lin(self._tokens(h.contiguous(memory_format=CL)), "quant_conv").view(B, Hh, Wh, C2).permute(0, 3, 1, 2)
        mean, logvar = m.float().chunk(2, dim=1)
        return mean.contiguous(), logvar.clamp(-30.0, 20.0).contiguous()

    def encode(self, x, noise=None, sample=True, scaled=True):
        """``vae.encode(x).latent_dist.sample() * scaling_factor`` (ppft_train.py:993-996); ``noise`` is the N(0,1) draw
        (injected so that runs are reproducible), ``sample=False`` returns the mode.  ``scaled=False`` leaves the 0.18215
        factor out, as stage 1 does (latent_wm_pretrain.py:171)."""
        mean, logvar = self.encode_moments(x)
        z = mean
        if sample:
            if noise is None:
                noise = torch.randn_like(mean)
            z = mean + torch.exp(0.5 * logvar) * noise.to(mean)
        return z * self.cfg["scaling_factor"] if scaled else z

    @torch.no_grad()
    def decode(self, z_scaled, scaled=True):
        """``vae.decode(z / scaling_factor).sample``: z [B,4,h,w] (already multiplied by 0.18215) -> image [B,3,8h,8w].
        ``scaled=False``: z is a raw posterior sample (``vae.decode(latents).sample`` of latent_wm_pretrain.py:100-104)."""
        return self.decode_grad(z_scaled, scaled)

    def decode_grad(self, z_scaled, scaled=True):
        """``decode`` with the autograd graph back to ``z_scaled`` kept: stage 1 trains the SecretEncoder THROUGH the frozen
        decoder (train/latent_wm_pretrain.py:180-181).  Every op's backward is a HIP kernel (conv backward-data with the
        upsample adjoint, GroupNorm backward, GEMMs, the GEMM-built wide-head attention); weights get no gradient."""
        rch, L_ = tuple(reversed(self.cfg["block_out_channels"])), self.cfg["layers_per_block"]
        z = z_scaled.to(self.device).float()
        z = (z / self.cfg["scaling_factor"] if scaled else z).to(torch.bfloat16)
        B, C, H, W = z.shape
        zp = torch.cat([z, z.new_zeros((B, 8 - C, H, W))], dim=1).contiguous(memory_format=CL)   # 1x1 conv: pad K to 8
        if "post_quant_conv8" not in self.p:
            pk = self.p["post_quant_conv"]
            wq = pk.w.new_zeros((8, 8))
            wq[:pk.N, :pk.K] = pk.w
            bq = pk.bias.new_zeros(8)
            bq[:pk.N] = pk.bias
            self.p["post_quant_conv8"] = ops.PackedLinear(wq, bq)
        h = self._lin(self._tokens(zp), "post_quant_conv8").view(B, H, W, 8).permute(0, 3, 1, 2)[:, :C]
        h = self._conv3(h.contiguous(memory_format=CL), "decoder.conv_in")
        h = self._mid(h, "decoder.mid_block")
        for i in range(len(rch)):
            for j in range(L_ + 1):
                h = self._resnet(h, f"decoder.up_blocks.{i}.resnets.{j}")
            if i + 1 < len(rch):
                h = self._conv3(h, f"decoder.up_blocks.{i}.upsamplers.0.conv", upsample=True)
        h = self._conv3(self._norm(h, "decoder.conv_norm_out", True), "decoder.conv_out")
        return h.float().contiguous()


def encode_gflop(cfg=SD15_VAE, size=512):
    """Algorithmic GFLOP of one image through the encoder (convs + attention + 1x1s), for bench roofline figures."""
    ch, L_ = cfg["block_out_channels"], cfg["layers_per_block"]
    fl, res, ci = 0.0, size, ch[0]
    fl += 2 * res * res * 9 * cfg["in_channels"] * ch[0]
    for i, co in enumerate(ch):
        for j in range(L_):
            fl += 2 * res * res * 9 * (ci * co + co * co) + (2 * res * res * ci * co if ci != co else 0)
            ci = co
        if i + 1 < len(ch):
            res //= 2
            fl += 2 * res * res * 9 * co * co
    n = res * res
    fl += 2 * (2 * n * 9 * ci * ci * 2) + 4 * 2 * n * ci * ci + 2 * 2 * n * n * ci
    fl += 2 * n * 9 * ci * 2 * cfg["latent_channels"]
    return fl / 1e9
