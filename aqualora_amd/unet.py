"""SD-1.5 ``UNet2DConditionModel`` host for the HIP kernels.

Same module tree / state-dict keys (686 tensors) and call surface as diffusers 0.24 (reference call sites
train/ppft_train.py:546-548, 1026-1035; architecture pinned through the in-tree twin
scripts/lib/original_unet.py:1311-1606), so a diffusers SD-1.5 checkpoint loads by key and
``unet(sample, t, ctx, cross_attention_kwargs={"scale": S}).sample`` behaves like the reference's patched U-Net.

MI355X-first choices:
  * activations stay bf16 channels-last from conv_in to conv_out; a feature map and its token view share memory;
  * nearest-x2 upsample is folded into the following conv's gather; the ResNet time-embedding add and every
    residual add are fused into GEMM/conv epilogues; GroupNorm+SiLU is one op;
  * every linear / 1x1 conv site goes through the fused LoRA GEMM, every 3x3 conv through the implicit-GEMM kernel.
``scale`` is threaded positionally to every LoRA-compatible host exactly like diffusers does.
"""
import math
from types import SimpleNamespace

import torch
import torch.nn as nn

from . import ops, synth
from .lora import (LoRACompatibleConv, LoRACompatibleLinear, _packed_conv3, _packed_linear)

SD15 = dict(in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
            cross_attention_dim=768, attention_heads=8, norm_groups=32,
            down_attn=(True, True, True, False), up_attn=(False, True, True, True))


class GroupNorm(nn.Module):
    def __init__(self, channels, eps, device=None, dtype=None):
        super().__init__()
        self.num_groups, self.num_channels, self.eps = 32, channels, eps
        self.weight = nn.Parameter(torch.ones(channels, device=device, dtype=dtype))
        self.bias = nn.Parameter(torch.zeros(channels, device=device, dtype=dtype))

    def forward(self, x, silu=False):
        return ops.groupnorm_silu(x, self.weight, self.bias, self.eps, silu)

    def forward_res(self, x, silu=False):
        """-> (norm(x), x): use the second value for the residual / shortcut branch (its gradient is then added inside the
        GroupNorm backward kernel)."""
        return ops.groupnorm_silu_res(x, self.weight, self.bias, self.eps, silu)


class LayerNorm(nn.Module):
    def __init__(self, channels, eps=1e-5, device=None, dtype=None):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(channels, device=device, dtype=dtype))
        self.bias = nn.Parameter(torch.zeros(channels, device=device, dtype=dtype))

    def forward(self, x):
        shp = x.shape
        return ops.layernorm(x.reshape(-1, shp[-1]), self.weight, self.bias, self.eps).view(shp)

    def forward_res(self, x):
        shp = x.shape
        y, xr = ops.layernorm_res(x.reshape(-1, shp[-1]), self.weight, self.bias, self.eps)
        return y.view(shp), xr.view(shp)


_FREQ_TABLES = {}


def timestep_freq_table(device, dim, downscale_freq_shift=0, max_period=10000):
    """[1, dim // 2] fp32 frequencies of the sinusoidal embedding (original_unet.py:340-347), built once per (device, dim,
    shift, period); a table built inside a graph capture lives in the capture's pool and is not cached."""
    half = dim // 2
    key = (str(device), dim, downscale_freq_shift, max_period)
    freq = _FREQ_TABLES.get(key)
    if freq is None:
        exponent = -math.log(max_period) * torch.arange(half, dtype=torch.float32, device=device)
        exponent = exponent / (half - downscale_freq_shift)
        freq = torch.exp(exponent)[None, :]
        if not (freq.is_cuda and torch.cuda.is_current_stream_capturing()):
            _FREQ_TABLES[key] = freq
    return freq


def get_timestep_embedding(timesteps, dim, flip_sin_to_cos=True, downscale_freq_shift=0, max_period=10000):
    """Sinusoidal embedding (original_unet.py:323-361), fp32.  The frequency table depends on the arguments only: built once per
    (device, dim, shift, period) instead of four launches per step; sin / cos are concatenated directly in the requested order."""
    freq = timestep_freq_table(timesteps.device, dim, downscale_freq_shift, max_period)
    emb = timesteps[:, None].float() * freq
    if flip_sin_to_cos:
        return torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1)
    return torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)


class TimestepEmbedding(nn.Module):
    def __init__(self, in_ch, dim, **kw):
        super().__init__()
        self.linear_1 = LoRACompatibleLinear(in_ch, dim, **kw)
        self.linear_2 = LoRACompatibleLinear(dim, dim, **kw)

    def forward(self, t_emb, scale=1.0):
        return self.linear_2(torch.nn.functional.silu(self.linear_1(t_emb, scale)), scale)


class ResnetBlock2D(nn.Module):
    def __init__(self, cin, cout, temb_dim, eps, **kw):
        super().__init__()
        self.norm1 = GroupNorm(cin, eps, **kw)
        self.conv1 = LoRACompatibleConv(cin, cout, 3, padding=1, **kw)
        self.time_emb_proj = LoRACompatibleLinear(temb_dim, cout, **kw)
        self.norm2 = GroupNorm(cout, eps, **kw)
        self.conv2 = LoRACompatibleConv(cout, cout, 3, padding=1, **kw)
        self.conv_shortcut = LoRACompatibleConv(cin, cout, 1, **kw) if cin != cout else None

    def forward(self, x, temb_act, scale=1.0):
        h, x = self.norm1.forward_res(x, silu=True)
        if isinstance(temb_act, dict):  # projections of all ResNets were batched into one GEMM (UNet forward)
            tproj = temb_act[id(self)]   # [B, cout] view with row stride = sum of all couts
        else:
            tproj = self.time_emb_proj(temb_act, scale).contiguous()
        # conv1's only consumer is norm2, called next: a split-K conv1 leaves its finalize launch to norm2's first pass; both
        # convolutions read a GroupNorm output nobody else reads: their backward-data launches leave theirs to the GroupNorm
        # backward kernels (ops.Conv3x3Fn, aql_conv3x3_*_defer / aql_groupnorm_silu_*_slabs)
        h = ops.conv3x3(h, _packed_conv3(self.conv1), False, tproj, None, gn_next=True, gn_input=True)
        h = self.norm2(h, silu=True)
        shortcut = x if self.conv_shortcut is None else self.conv_shortcut(x, scale)
        return ops.conv3x3(h, _packed_conv3(self.conv2), False, None, shortcut, gn_input=True)


class Downsample2D(nn.Module):
    def __init__(self, ch, **kw):
        super().__init__()
        self.conv = LoRACompatibleConv(ch, ch, 3, stride=2, padding=1, **kw)

    def forward(self, x, scale=1.0):
        return self.conv(x, scale)


class Upsample2D(nn.Module):
    def __init__(self, ch, **kw):
        super().__init__()
        self.conv = LoRACompatibleConv(ch, ch, 3, padding=1, **kw)

    def forward(self, x, scale=1.0):
        return ops.conv3x3(x, _packed_conv3(self.conv), True, None, None)


def _has_alpha(host):
    """`up_hidden_states *= network_alpha / rank` (lora_modules.py:21-22, 39-40) is folded into the scale by lora._run_linear
    only: the grouped q|k|v, grouped text-state k|v and fused feed-forward launches take S as it is, so a site that carries a
    network_alpha sends its host down the per-site path.  (Always None on the PPFT path, ppft_train.py:662-666.)"""
    ll = getattr(host, "lora_layer", None)
    return ll is not None and getattr(ll, "network_alpha", None) is not None


class Attention(nn.Module):
    def __init__(self, query_dim, cross_dim, heads, **kw):
        super().__init__()
        self.heads = heads
        cross_dim = cross_dim if cross_dim is not None else query_dim
        self.to_q = LoRACompatibleLinear(query_dim, query_dim, bias=False, **kw)
        self.to_k = LoRACompatibleLinear(cross_dim, query_dim, bias=False, **kw)
        self.to_v = LoRACompatibleLinear(cross_dim, query_dim, bias=False, **kw)
        self.to_out = nn.ModuleList([LoRACompatibleLinear(query_dim, query_dim, **kw), nn.Dropout(0.0)])

    def _fused_weights(self, cross):
        """Concatenated projection weights for the LoRA-free passes: [Wq;Wk;Wv] (self-attention) or [Wk;Wv] (cross).  Rebuilt
        whenever a projection's packed copy is replaced (weights re-initialised or a LoRA fused into them)."""
        pks = tuple(_packed_linear(m) for m in (self.to_q, self.to_k, self.to_v))
        c = getattr(self, "_aql_qkv", None)
        if c is None or c[0] is not pks[0] or c[1] is not pks[1] or c[2] is not pks[2]:
            w = torch.cat([p.w for p in (pks[1:] if cross else pks)], dim=0).contiguous()
            c = pks + (w,)
            object.__setattr__(self, "_aql_qkv", c)
        return c[3]

    def _fused_weights_t(self):
        """[Wq^T | Wk^T | Wv^T] ([K, 3C], column blocks): the weight-side operand of dX = [dQ | dK | dV].[Wq | Wk | Wv] in the grouped
        backward of the self-attention projections (ops.GroupedWideFn).  Rebuilt with the packed copies, like _fused_weights."""
        return ops.wcat_t(tuple(_packed_linear(m) for m in (self.to_q, self.to_k, self.to_v)))

    def _forward_nolora(self, x, ctx, residual):
        """The frozen 'clean' pass and inference (scale None, no autograd): q|k|v come from ONE GEMM (two launches fewer per
        self-attention, one fewer per cross-attention); the attention kernels read the packed result through row strides.
        No split-K at these depths, so every element is accumulated in the same order as by the three separate GEMMs:
        bit-identical (tests/test_gpu_parity.py compares the two paths with torch.equal)."""
        B, N, C = x.shape
        x2d = x.reshape(B * N, C)
        if ctx is None:
            qkv = ops.gemm_bf16(x2d, self._fused_weights(False)).view(B, N, 3 * C)
            q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
        else:
            q = self.to_q(x, None)
            cache = getattr(ctx, "_aql_kv_static", None)   # a sampling loop computed them once per prompt (UNet.text_kv)
            k, v = cache[id(self)] if cache is not None and id(self) in cache else self._text_kv(ctx)
        o = ops.attention(q, k, v, self.heads)
        return self.to_out[0](o, None, residual=residual)

    def _text_kv(self, ctx):
        Bc, Nc, Cc = ctx.shape
        C = self.to_q.out_features
        c2d = ctx.reshape(Bc * Nc, Cc)
        if c2d.dtype != torch.bfloat16:
            c2d = c2d.to(torch.bfloat16)
        kv = ops.gemm_bf16(c2d.contiguous(), self._fused_weights(True)).view(Bc, Nc, 2 * C)
        return kv[..., :C], kv[..., C:]

    def _grouped_qkv(self, hidden_states, scale):
        """q|k|v of a self-attention as ONE grouped LoRA launch (they read the same tokens); None when the grouped form does
        not apply (no LoRA, rank != 32, float scale, bf16 copies not stacked by a LoraBank ...)."""
        if not torch.is_tensor(scale) or hidden_states.dim() != 3 or hidden_states.dtype != torch.bfloat16:
            return None
        mods = (self.to_q, self.to_k, self.to_v)
        if any(m.lora_layer is None or m.bias is not None or _has_alpha(m) for m in mods):
            return None
        from .lora import _scale16, _site_of
        sites = [_site_of(m.lora_layer) for m in mods]
        packs = [_packed_linear(m) for m in mods]
        B, N, C = hidden_states.shape
        x2d = hidden_states.reshape(B * N, C)
        r = sites[0].rank
        if any(st.rank != r for st in sites) or scale.dim() != 2 or scale.shape[1] != r:
            return None
        S = _scale16(scale, B, r, x2d.device)
        S16 = getattr(S, "_aql_s16", None)
        if S16 is None:
            S16 = S.detach().to(torch.bfloat16).contiguous()
            S._aql_s16 = S16
        if not x2d.is_contiguous():
            return None
        if r > 32:
            # rank 320 (BASELINE config 3): four launches for the three projections and their backward (ops.GroupedWideFn); training runs
            # need the trainer's deferred weight-gradient / dS machinery, forward-only runs (sampling through the un-fused LoRA) nothing
            train = torch.is_grad_enabled()
            if (train and (ops.DEFERRED is None or getattr(S, "_aql_ds_accum", None) is None)) or not ops.grouped_wide_ok(x2d, packs, sites, S16):
                return None
            q, k, v = ops.lora_linear_grouped_wide(x2d, self._fused_weights(False), self._fused_weights_t(), packs, sites, S, S16, N)
            return q.view(B, N, C), k.view(B, N, C), v.view(B, N, C), 1     # (1: the attention backward packs [dQ | dK | dV])
        if not ops.grouped_lora_ok(x2d, packs, sites, S16):
            return None
        q, k, v = ops.lora_linear_grouped(x2d, self._fused_weights(False), packs, sites, S, S16, N)
        return q.view(B, N, C), k.view(B, N, C), v.view(B, N, C)

    def _grouped_kv_wide(self, ctx, scale):
        """k | v of a text-state attention at a LoRA rank above 32 as one grouped pair (ops.GroupedWideFn, G = 2; the text states carry
        no gradient: three launches forward, one backward, where the two sites run four and two).  -> (k, v) or None."""
        if not torch.is_tensor(scale) or ctx.dim() != 3 or ctx.dtype != torch.bfloat16 or scale.dim() != 2 or not ctx.is_contiguous():
            return None
        mods = (self.to_k, self.to_v)
        if any(m.lora_layer is None or m.bias is not None or _has_alpha(m) for m in mods):
            return None
        from .lora import _scale16, _site_of
        sites = [_site_of(m.lora_layer) for m in mods]
        r = sites[0].rank
        if r <= 32 or scale.shape[1] != r or ctx.requires_grad:
            return None
        packs = [_packed_linear(m) for m in mods]
        B, Nc, Cc = ctx.shape
        x2d = ctx.reshape(B * Nc, Cc)
        S = _scale16(scale, B, r, x2d.device)
        S16 = getattr(S, "_aql_s16", None)
        if S16 is None:
            S16 = S.detach().to(torch.bfloat16).contiguous()
            S._aql_s16 = S16
        train = torch.is_grad_enabled()
        if (train and (ops.DEFERRED is None or getattr(S, "_aql_ds_accum", None) is None)) or \
                not ops.grouped_wide_ok(x2d, packs, sites, S16, need_dx=False):
            return None
        k, v = ops.lora_linear_grouped_wide(x2d, self._fused_weights(True), None, packs, sites, S, S16, Nc)
        C = packs[0].N
        return k.view(B, Nc, C), v.view(B, Nc, C)

    def forward(self, hidden_states, encoder_hidden_states=None, scale=1.0, residual=None):
        if (scale is None and not torch.is_grad_enabled() and hidden_states.dim() == 3
                and hidden_states.dtype == torch.bfloat16 and hidden_states.is_contiguous()):
            return self._forward_nolora(hidden_states, encoder_hidden_states, residual)
        qkv = None
        if encoder_hidden_states is None:
            qkv = self._grouped_qkv(hidden_states, scale)
        else:
            cache = getattr(encoder_hidden_states, "_aql_kv", None)   # k|v of all cross-attentions, one launch (UNet.forward)
            if cache is not None and id(self) in cache:
                k, v = cache[id(self)]
                qkv = (self.to_q(hidden_states, scale), k, v)
            else:
                kv = self._grouped_kv_wide(encoder_hidden_states, scale)
                if kv is not None:
                    qkv = (self.to_q(hidden_states, scale), kv[0], kv[1], 2)   # (2: the attention backward packs [dK | dV])
        if qkv is None:
            ctx = hidden_states if encoder_hidden_states is None else encoder_hidden_states
            qkv = ops.parallel([lambda: self.to_q(hidden_states, scale), lambda: self.to_k(ctx, scale),
                                lambda: self.to_v(ctx, scale)])
        pack = qkv[3] if len(qkv) == 4 else 0      # the grouped wide-rank projections read their output gradients as one buffer
        q, k, v = qkv[:3]
        o = ops.attention(q, k, v, self.heads, pack_grads=pack)
        return self.to_out[0](o, scale, residual=residual)


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out, **kw):
        super().__init__()
        self.proj = LoRACompatibleLinear(dim_in, dim_out * 2, **kw)

    def forward(self, x, scale=1.0):
        # proj + LoRA + value * gelu(gate) in ONE launch: the activation runs in the GEMM epilogue (original_unet.py:727-729)
        return self.proj(x, scale, geglu=True)


class FeedForward(nn.Module):
    def __init__(self, dim, **kw):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * 4, **kw), nn.Dropout(0.0), LoRACompatibleLinear(dim * 4, dim, **kw)])

    def forward(self, x, scale=1.0, residual=None):
        y = self._fused(x, scale, residual)
        if y is not None:
            return y
        return self.net[2](self.net[0](x, scale), scale, residual=residual)

    def _fused(self, x, scale, residual):
        """Training with the watermark LoRA on both linears: the whole feed-forward as one autograd node (ops.FeedForwardFn)."""
        p0, p2 = self.net[0].proj, self.net[2]
        if (not torch.is_grad_enabled() or not torch.is_tensor(scale) or p0.lora_layer is None or p2.lora_layer is None
                or x.dim() != 3 or x.dtype != torch.bfloat16 or ops.REF_ROUNDING or _has_alpha(p0) or _has_alpha(p2)):
            return None
        from .lora import _scale16, _site_of
        s0, s2 = _site_of(p0.lora_layer), _site_of(p2.lora_layer)
        if s0.rank != s2.rank or scale.dim() != 2 or scale.shape[1] != s0.rank:
            return None
        B, N, C = x.shape
        x2d = x.reshape(B * N, C)
        if not x2d.is_contiguous():
            return None
        S = _scale16(scale, B, s0.rank, x2d.device)
        S16 = getattr(S, "_aql_s16", None)
        if S16 is None:
            S16 = S.detach().to(torch.bfloat16).contiguous()
            S._aql_s16 = S16
        res2d = None if residual is None else residual.reshape(B * N, -1).contiguous()
        y = ops.feed_forward(x2d, _packed_linear(p0), s0, _packed_linear(p2), s2, S, S16, N, res2d)
        return y.view(B, N, -1)


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, cross_dim, **kw):
        super().__init__()
        self.norm1 = LayerNorm(dim, **kw)
        self.attn1 = Attention(dim, None, heads, **kw)
        self.norm2 = LayerNorm(dim, **kw)
        self.attn2 = Attention(dim, cross_dim, heads, **kw)
        self.norm3 = LayerNorm(dim, **kw)
        self.ff = FeedForward(dim, **kw)

    def forward(self, h, ctx, scale=1.0):
        n, h = self.norm1.forward_res(h)
        h = self.attn1(n, None, scale, residual=h)
        n, h = self.norm2.forward_res(h)
        h = self.attn2(n, ctx, scale, residual=h)
        n, h = self.norm3.forward_res(h)
        return self.ff(n, scale, residual=h)


class Transformer2DModel(nn.Module):
    def __init__(self, ch, heads, cross_dim, **kw):
        super().__init__()
        self.norm = GroupNorm(ch, 1e-6, **kw)
        self.proj_in = LoRACompatibleConv(ch, ch, 1, **kw)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(ch, heads, cross_dim, **kw)])
        self.proj_out = LoRACompatibleConv(ch, ch, 1, **kw)

    def forward(self, x, ctx, scale=1.0):
        B, C, H, W = x.shape
        n, x = self.norm.forward_res(x)
        y = self._forward_chains(n, x, ctx, scale)
        if y is not None:
            return y
        h = self.proj_in(n, scale)
        tokens = ops.nhwc_view(h).reshape(B, H * W, C)
        for blk in self.transformer_blocks:
            tokens = blk(tokens, ctx, scale)
        h = tokens.view(B, H, W, C).permute(0, 3, 1, 2)
        return self.proj_out(h, scale, residual=x)

    def _forward_chains(self, n, x, ctx, scale):
        """The 320-channel level with the rank-32 / rank-320 watermark LoRA on every linear (training, twin or plain batch): the linears with
        K = N = 320 and the row-local operations between them run as row-resident chains (ops.ChainFn, csrc/aql_chain.hip) --
            proj_in -> norm1 -> to_q | to_k | to_v      |  attn1.to_out + residual -> norm2 -> attn2.to_q
            attn2.to_out + residual -> norm3            |  (feed-forward and proj_out: the existing launches)
        13 launches of the block become 8; every tensor has the bits of the per-launch path (tests/test_gpu_kernels.py).  Returns
        None when the chain form does not apply (other widths, ranks, float scale, no LoRA, small batches: ops.chain_ok)."""
        nolora = scale is None and not torch.is_grad_enabled()     # the clean pass alone / sampling with the LoRA fused into W
        if not ops.CHAIN or len(self.transformer_blocks) != 1 or n.dtype != torch.bfloat16 or n.shape[1] != 320:
            return None
        if not nolora and (not torch.is_tensor(scale) or scale.dim() != 2 or scale.shape[1] not in (32, 320)):
            return None
        blk = self.transformer_blocks[0]
        a1, a2 = blk.attn1, blk.attn2
        hosts = (self.proj_in, a1.to_q, a1.to_k, a1.to_v, a1.to_out[0], a2.to_q, a2.to_out[0])
        if any(m.bias is not None for m in (a1.to_q, a1.to_k, a1.to_v, a2.to_q)):
            return None
        if not nolora and any(m.lora_layer is None or _has_alpha(m) for m in hosts):
            return None
        B, C, H, W = n.shape
        N = H * W
        if nolora:
            kvs = getattr(ctx, "_aql_kv_static", None)      # a sampling loop computed them once per prompt (UNet.text_kv)
            kv = None if kvs is None or id(a2) not in kvs else kvs
        else:
            kv = getattr(ctx, "_aql_kv", None)   # k | v of all cross-attentions, computed in front of the U-Net (UNet._ctx_kv)
            if kv is not None and id(a2) not in kv:
                return None
        from .lora import _scale16, _site_of
        n = ops.as_cl(n)
        x2d = ops.nhwc_view(n).reshape(B * N, C)
        pk = {m: _packed_linear(m) for m in hosts}
        st = {m: (None if nolora else _site_of(m.lora_layer)) for m in hosts}
        S = S16 = None
        if not nolora:
            S = _scale16(scale, B, scale.shape[1], x2d.device)
            S16 = getattr(S, "_aql_s16", None)
            if S16 is None:
                S16 = S.detach().to(torch.bfloat16).contiguous()
                S._aql_s16 = S16
        CS = ops.ChainStage
        # attn1.to_q leaves the chain multiplied by d^-1/2 log2(e) (one bf16 rounding, as the unscaled q has): the self-attention forward
        # then carries its softmax shift inside the S-product (aql_sdpa_fwd_qpre).  Head sizes with a spare column only (d = 40 here).
        d_head = C // a1.heads
        qpre = ops.QPRE and d_head < 64 and d_head % 8 == 0
        qs = (d_head ** -0.5) * 1.4426950408889634 if qpre else 1.0
        d_stages = [CS(pk[self.proj_in], st[self.proj_in], True, ln=blk.norm1), CS(pk[a1.to_q], st[a1.to_q], False, oscale=qs)] + \
                   [CS(pk[m], st[m], False) for m in (a1.to_k, a1.to_v)]
        a_stages = [CS(pk[a1.to_out[0]], st[a1.to_out[0]], True, use_res=True, ln=blk.norm2), CS(pk[a2.to_q], st[a2.to_q], False)]
        c_stages = [CS(pk[a2.to_out[0]], st[a2.to_out[0]], True, use_res=True, ln=blk.norm3, emit_n=True)]
        # all seven hosts are gated BEFORE the first chain is launched (shape, rank, the opt-in weight-side form): the later chains read
        # o1 / o2, which have x2d's geometry (twin views of [2B N, C] buffers, row stride C)
        if not ops.chain_ok(x2d, d_stages + a_stages + c_stages, S16, N):
            return None
        h0, q, k, v = ops.lora_chain(x2d, None, S, S16, N, d_stages)
        # (rank > 32: the attention backward writes [dQ | dK | dV] as one buffer for the grouped backward of the three DIRECT stages)
        pack1 = 1 if (S16 is not None and S16.shape[1] > 32 and ops.GROUPED_WIDE) else 0
        o1 = ops.attention(q.view(B, N, C), k.view(B, N, C), v.view(B, N, C), a1.heads, q_prescaled=qpre, pack_grads=pack1).reshape(B * N, C)
        if not ops.chain_input_ok(o1, x2d):
            raise ops.L.AqlError("Transformer2DModel chains: the self-attention output lost the twin geometry of its input")
        h1, q2 = ops.lora_chain(o1, h0, S, S16, N, a_stages)
        pack2 = 0
        if kv is not None:
            k2, v2 = kv[id(a2)]
        elif nolora:
            k2, v2 = a2._text_kv(ctx)
        else:                                   # (rank 320: no grouped k | v launch in front of the U-Net; the pair of this block as one group)
            kvw = a2._grouped_kv_wide(ctx, scale)
            pack2 = 2 if kvw is not None else 0
            k2, v2 = kvw if kvw is not None else (a2.to_k(ctx, scale), a2.to_v(ctx, scale))
        o2 = ops.attention(q2.view(B, N, C), k2, v2, a2.heads, pack_grads=pack2).reshape(B * N, C)
        if not ops.chain_input_ok(o2, x2d):
            raise ops.L.AqlError("Transformer2DModel chains: the text-state attention output lost the twin geometry of its input")
        h2, n3 = ops.lora_chain(o2, h1, S, S16, N, c_stages)
        tokens = blk.ff(n3.view(B, N, C), scale, residual=h2.view(B, N, C))
        h = tokens.view(B, H, W, C).permute(0, 3, 1, 2)
        return self.proj_out(h, scale, residual=x)


class DownBlock(nn.Module):
    def __init__(self, cin, cout, temb_dim, n_layers, attn, heads, cross_dim, add_down, eps, **kw):
        super().__init__()
        self.has_cross_attention = attn
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, temb_dim, eps, **kw)
                                      for i in range(n_layers)])
        if attn:
            self.attentions = nn.ModuleList([Transformer2DModel(cout, heads, cross_dim, **kw) for _ in range(n_layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(cout, **kw)]) if add_down else None

    def forward(self, h, temb_act, ctx, scale):
        outs = ()
        for i, res in enumerate(self.resnets):
            h = res(h, temb_act, scale)
            if self.has_cross_attention:
                h = self.attentions[i](h, ctx, scale)
            outs += (h,)
        if self.downsamplers is not None:
            h = self.downsamplers[0](h, scale)
            outs += (h,)
        return h, outs


class MidBlock(nn.Module):
    def __init__(self, ch, temb_dim, heads, cross_dim, eps, **kw):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(ch, ch, temb_dim, eps, **kw) for _ in range(2)])
        self.attentions = nn.ModuleList([Transformer2DModel(ch, heads, cross_dim, **kw)])

    def forward(self, h, temb_act, ctx, scale):
        h = self.resnets[0](h, temb_act, scale)
        h = self.attentions[0](h, ctx, scale)
        return self.resnets[1](h, temb_act, scale)


class UpBlock(nn.Module):
    def __init__(self, cin, cout, prev, temb_dim, n_layers, attn, heads, cross_dim, add_up, eps, **kw):
        super().__init__()
        self.has_cross_attention = attn
        resnets = []
        for i in range(n_layers):
            skip = cin if i == n_layers - 1 else cout
            rin = prev if i == 0 else cout
            resnets.append(ResnetBlock2D(rin + skip, cout, temb_dim, eps, **kw))
        self.resnets = nn.ModuleList(resnets)
        if attn:
            self.attentions = nn.ModuleList([Transformer2DModel(cout, heads, cross_dim, **kw) for _ in range(n_layers)])
        self.upsamplers = nn.ModuleList([Upsample2D(cout, **kw)]) if add_up else None

    def forward(self, h, skips, temb_act, ctx, scale):
        for i, res in enumerate(self.resnets):
            h = ops.cat_channels(h, skips[-1 - i])
            h = res(h, temb_act, scale)
            if self.has_cross_attention:
                h = self.attentions[i](h, ctx, scale)
        if self.upsamplers is not None:
            h = self.upsamplers[0](h, scale)
        return h


class UNet2DConditionModel(nn.Module):
    def __init__(self, config=None, device=None, dtype=torch.bfloat16):
        super().__init__()
        cfg = dict(SD15)
        cfg.update(config or {})
        self.config = SimpleNamespace(**cfg)
        kw = dict(device=device, dtype=dtype)
        boc = cfg["block_out_channels"]
        L = cfg["layers_per_block"]
        heads, cross = cfg["attention_heads"], cfg["cross_attention_dim"]
        temb_dim = boc[0] * 4
        eps = 1e-5
        self.conv_in = LoRACompatibleConv(cfg["in_channels"], boc[0], 3, padding=1, **kw)
        self.time_embedding = TimestepEmbedding(boc[0], temb_dim, **kw)
        downs = []
        out_ch = boc[0]
        for i, ch in enumerate(boc):
            in_ch, out_ch = out_ch, ch
            downs.append(DownBlock(in_ch, out_ch, temb_dim, L, cfg["down_attn"][i], heads, cross, i != len(boc) - 1,
                                   eps, **kw))
        self.down_blocks = nn.ModuleList(downs)
        self.mid_block = MidBlock(boc[-1], temb_dim, heads, cross, eps, **kw)
        ups = []
        rev = list(reversed(boc))
        out_ch = rev[0]
        for i in range(len(boc)):
            prev = out_ch
            out_ch = rev[i]
            in_ch = rev[min(i + 1, len(boc) - 1)]
            ups.append(UpBlock(in_ch, out_ch, prev, temb_dim, L + 1, cfg["up_attn"][i], heads, cross,
                               i != len(boc) - 1, eps, **kw))
        self.up_blocks = nn.ModuleList(ups)
        self.conv_norm_out = GroupNorm(boc[0], eps, **kw)
        self.conv_out = LoRACompatibleConv(boc[0], cfg["out_channels"], 3, padding=1, **kw)
        for p in self.parameters():
            p.requires_grad_(False)

    def _all_time_projections(self, temb_act):
        """time_emb_proj(SiLU(temb)) of all 22 ResNets (original_unet.py:449) as ONE GEMM: they depend only on the
        timestep, so 22 tiny launches per pass collapse into one; each ResNet reads its [B, cout] column slice."""
        cat = getattr(self, "_aql_temb_cat", None)
        if cat is None:
            blocks = [m for m in self.modules() if isinstance(m, ResnetBlock2D)]
            w = torch.cat([m.time_emb_proj.weight.detach() for m in blocks], dim=0)
            b = torch.cat([m.time_emb_proj.bias.detach() for m in blocks], dim=0)
            offs, o = [], 0
            for m in blocks:
                offs.append((id(m), o, m.time_emb_proj.out_features))
                o += m.time_emb_proj.out_features
            cat = (ops.PackedLinear(w, b), offs)
            object.__setattr__(self, "_aql_temb_cat", cat)
        packed, offs = cat
        allp = ops.lora_linear(temb_act.contiguous(), packed)
        return {key: allp[:, o:o + n] for key, o, n in offs}, allp

    def _time_projection_views(self, allp):
        """Per-ResNet column views of a [rows, sum cout] block of time projections (a row of `time_projection_rows`, gathered
        per step by a sampling loop: `forward(..., _aql_tproj=...)`)."""
        return {key: allp[:, o:o + n] for key, o, n in self._aql_temb_cat[1]}

    @torch.no_grad()
    def time_projection_rows(self, timestep, rows, scale=None):
        """The timestep-only head of `forward` on its own: sinusoid -> time_embedding -> SiLU -> the 22 time_emb_proj as one
        GEMM, for `rows` samples that share `timestep`; returns the [rows, sum cout] block.  A sampling loop calls this once
        per step of its schedule BEFORE the loop (same launches on the same shapes as inside `forward`: same bits) and feeds
        the row of the current step back through `_aql_tproj` -- 19 element-wise / tiny-GEMM launches leave every step."""
        if not torch.is_tensor(timestep):
            timestep = torch.tensor([timestep], device=self.device)
        timestep = timestep.reshape(-1).expand(rows)
        t_emb = get_timestep_embedding(timestep, self.config.block_out_channels[0]).to(self.dtype)
        emb = self.time_embedding(t_emb, scale)
        return self._all_time_projections(torch.nn.functional.silu(emb))[1]

    @torch.no_grad()
    def text_kv(self, ctx):
        """attn2's k|v projections of the text states for the LoRA-free / fused-LoRA passes (`Attention._forward_nolora`): they do
        not depend on the latents, so a sampling loop computes them ONCE per prompt (same GEMM as inside the block: same bits)
        and attaches the dict to the text-state tensor as ``_aql_kv_static``; 16 launches leave every step."""
        out = {}
        for m in self.modules():
            if isinstance(m, BasicTransformerBlock):
                out[id(m.attn2)] = m.attn2._text_kv(ctx)
        return out

    def _ctx_kv(self, ctx, scale):
        """attn2.to_k / to_v of ALL cross-attentions depend only on the text states: 32 small LoRA linears (616 x 768 -> C at
        twin batch 8) become ONE grouped launch in front of the U-Net; every cross-attention picks its k, v column views from
        ``ctx._aql_kv``.  Skipped (the per-site path runs) unless the grouped form applies."""
        if hasattr(ctx, "_aql_kv"):
            del ctx._aql_kv
        if not torch.is_tensor(scale) or ctx.dim() != 3 or scale.dim() != 2 or scale.shape[1] != 32:
            return
        from .lora import _scale16, _site_of
        cache = getattr(self, "_aql_ctxkv", None)
        if cache is None:
            attns = [m.attn2 for m in self.modules() if isinstance(m, BasicTransformerBlock)]
            names = {id(m): n for n, m in self.named_modules()}
            attns.sort(key=lambda a: names[id(a)])
            mods = [m for a in attns for m in (a.to_k, a.to_v)]
            cache = (attns, mods)
            object.__setattr__(self, "_aql_ctxkv", cache)
        attns, mods = cache
        if not mods or any(m.lora_layer is None or m.bias is not None or _has_alpha(m) for m in mods):
            return
        sites = [_site_of(m.lora_layer) for m in mods]
        packs = [_packed_linear(m) for m in mods]
        if any(st.rank != 32 for st in sites):
            return
        B, N, C = ctx.shape
        x2d = ctx.reshape(B * N, C)
        S = _scale16(scale, B, 32, x2d.device)
        S16 = getattr(S, "_aql_s16", None)
        if S16 is None:
            S16 = S.detach().to(torch.bfloat16).contiguous()
            S._aql_s16 = S16
        if not ops.grouped_lora_ok(x2d, packs, sites, S16):
            return
        wc = getattr(self, "_aql_ctxkv_w", None)
        if wc is None or any(a is not b for a, b in zip(wc[0], packs)):
            wc = (packs, torch.cat([p.w for p in packs], dim=0).contiguous())
            object.__setattr__(self, "_aql_ctxkv_w", wc)
        outs = ops.lora_linear_grouped(x2d, wc[1], packs, sites, S, S16, N)
        ctx._aql_kv = {id(a): (outs[2 * i].view(B, N, -1), outs[2 * i + 1].view(B, N, -1)) for i, a in enumerate(attns)}

    def __getstate__(self):
        """copy.deepcopy / pickle: captured sampling loops (inference._GuidedLoop) are derived state; a HIP graph cannot be copied."""
        state = dict(self.__dict__)
        state.pop("_aql_loops", None)
        return state

    @property
    def dtype(self):
        return self.conv_in.weight.dtype

    @property
    def device(self):
        return self.conv_in.weight.device

    def forward(self, sample, timestep, encoder_hidden_states, class_labels=None, cross_attention_kwargs=None,
                return_dict=True, _aql_t_emb=None, _aql_tproj=None):
        """``_aql_t_emb`` (internal): the sinusoidal timestep embedding of every row of the (twin) batch, already built by the
        trainer's prologue kernel (ppft.PPFTTrainer._twin_prologue) -- the eight element-wise launches of the generic path are skipped.
        ``_aql_tproj`` (internal): the [rows, sum cout] time projections of this step, computed before a sampling loop by
        `time_projection_rows`; the whole timestep head is skipped."""
        scale = 1.0
        if cross_attention_kwargs is not None and "scale" in cross_attention_kwargs:
            scale = cross_attention_kwargs["scale"]
        if not torch.is_tensor(timestep):
            timestep = torch.tensor([timestep], device=sample.device)
        timestep = timestep.reshape(-1).expand(sample.shape[0])
        # a channel-padded twin view (ops.register_cpad) passes through only at the width conv_in's packed weights expect --
        # the same test Conv3x3Fn makes; any other layout is made channels-last here (and re-registered by make_twin's callers)
        if not (sample.dtype == self.dtype and ops.is_cpad(sample, _packed_conv3(self.conv_in).Cin)):
            sample = ops.as_cl(sample.to(self.dtype))
        if _aql_tproj is not None:
            temb_act = self._time_projection_views(_aql_tproj)
        else:
            if _aql_t_emb is not None:
                t_emb = _aql_t_emb
            else:
                if ops._full(sample) is not None:
                    # twin batch (ops._Dual): `sample` is the watermarked half of a 2B buffer whose first half is the clean pass; both
                    # halves share the timesteps, and the per-ResNet time projections are per-sample row biases covering all 2B rows
                    timestep = timestep.repeat(2)
                t_emb = get_timestep_embedding(timestep, self.config.block_out_channels[0]).to(self.dtype)
            emb = self.time_embedding(t_emb, scale)
            temb_act = torch.nn.functional.silu(emb)  # every ResNet applies SiLU to temb first (original_unet.py:449)
            temb_act = self._all_time_projections(temb_act)[0]
        ctx = encoder_hidden_states.to(self.dtype).contiguous()
        self._ctx_kv(ctx, scale)
        try:
            h = self.conv_in(sample, scale)
            skips = (h,)
            # backward-leg hooks of the data-parallel trainer (ppft.PPFTTrainer, lora.backward_stage): hooks[0] fires when backward
            # has left the up path (the gradient of the mid-block output is complete), hooks[1] after the mid block and
            # down_blocks.3 / .2 (gradient of down_blocks.1's output: its skip-connection consumers in the up path and
            # down_blocks.2 are all done), hooks[2] after down_blocks.1 (gradient of down_blocks.0's output)
            hooks = getattr(self, "_aql_bwd_hooks", None)

            def leg_done(tensor, k):
                if hooks is not None and k < len(hooks) and hooks[k] is not None and tensor.requires_grad:
                    tensor.register_hook(lambda g, _f=hooks[k]: (_f(), None)[1])

            for bi, blk in enumerate(self.down_blocks):
                h, outs = blk(h, temb_act, ctx, scale)
                skips += outs
                if bi == 0:
                    leg_done(h, 2)
                elif bi == 1:
                    leg_done(h, 1)
            h = self.mid_block(h, temb_act, ctx, scale)
            leg_done(h, 0)
            for blk in self.up_blocks:
                n = len(blk.resnets)
                h = blk(h, skips[-n:], temb_act, ctx, scale)
                skips = skips[:-n]
        finally:
            # `ctx` can be the caller's own tensor (bf16, contiguous): the grouped k|v side channel must not outlive this
            # forward -- it would pin the [M, sum N] buffer between steps and hand stale k, v to a later direct block call
            if hasattr(ctx, "_aql_kv"):
                del ctx._aql_kv
        h = self.conv_norm_out(h, silu=True)
        h = self.conv_out(h, scale)
        if not return_dict:
            return (h,)
        return SimpleNamespace(sample=h)


def lora_keys(unet):
    """The 192 injection points of utils/unet_keys.json, derived from the module tree: every proj_in / proj_out /
    attn{1,2}.to_{q,k,v,out.0} / ff.net.0.proj / ff.net.2 of every transformer block, in sorted order."""
    keys = []
    for name, m in unet.named_modules():
        if isinstance(m, Transformer2DModel):
            keys += [f"{name}.proj_in", f"{name}.proj_out"]
            for j in range(len(m.transformer_blocks)):
                tb = f"{name}.transformer_blocks.{j}"
                for a in ("attn1", "attn2"):
                    keys += [f"{tb}.{a}.to_k", f"{tb}.{a}.to_out.0", f"{tb}.{a}.to_q", f"{tb}.{a}.to_v"]
                keys += [f"{tb}.ff.net.0.proj", f"{tb}.ff.net.2"]
    return sorted(keys)


@torch.no_grad()
def init_synthetic(unet, seed=2048):
    """Counter-based synthetic weights: std 1/sqrt(fan_in), norm gamma 1 / beta 0, bias std 0.02."""
    for name, p in unet.named_parameters():
        if "lora_layer" in name:
            continue
        if name.endswith("weight") and p.dim() >= 2:
            fan_in = p[0].numel()
            p.copy_(synth.normal(name, p.shape, fan_in ** -0.5, seed, p.device).to(p.dtype))
        elif name.endswith("weight"):
            p.fill_(1.0)
        elif "norm" in name:
            p.zero_()
        else:
            p.copy_(synth.normal(name, p.shape, 0.02, seed, p.device).to(p.dtype))
        mod = unet.get_submodule(name.rsplit(".", 1)[0])
        if hasattr(mod, "_aql_packed"):
            object.__delattr__(mod, "_aql_packed")
    if hasattr(unet, "_aql_temb_cat"):
        object.__delattr__(unet, "_aql_temb_cat")
