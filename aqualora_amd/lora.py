"""The watermark-LoRA plug-in surface, MI355X-native.

Mirrors, name for name, what the reference monkey-patches onto diffusers 0.24 modules:

  reference (utils/lora_modules.py)              here (HIP-backed, same signature and semantics)
  CustomLoRALinearLayerforward        :9-26      CustomLoRALinearLayerforward
  CustomLoRAConv2dLayerforward        :28-44     CustomLoRAConv2dLayerforward
  CustomLoRACompatibleConvforward     :46-54     CustomLoRACompatibleConvforward
  CustomLoRACompatibleLinearforward   :56-62     CustomLoRACompatibleLinearforward

plus light-weight twins of the diffusers host / LoRA layer classes (``LoRACompatibleLinear``,
``LoRACompatibleConv``, ``LoRALinearLayer``, ``LoRAConv2dLayer``; diffusers is not installed on the target
image) and the injection / patch loops of train/ppft_train.py:620-689 (``inject_lora``, ``patch_lora_forwards``).

``scale`` follows the reference: a float multiplies the LoRA branch, a ``[B, r]`` tensor is the per-message
diagonal S (``B.diag(S).A``).  ``scale=None`` is an extension: skip the LoRA branch (bit-identical to the
reference's all-zero scale of the "clean" pass, ppft_train.py:1026-1029, without spending the FLOPs).
The fused host forwards also accept ``residual=`` (added in the GEMM epilogue).
"""
import types

import torch
import torch.nn as nn

from . import ops


# ----------------------------------------------------------------------------- diffusers-compatible classes
class LoRALinearLayer(nn.Module):
    """down: Linear(in, rank, bias=False) ~ N(0, 1/rank); up: Linear(rank, out, bias=False) = 0 (diffusers 0.24)."""

    def __init__(self, in_features, out_features, rank=4, network_alpha=None, device=None, dtype=None):
        super().__init__()
        self.down = nn.Linear(in_features, rank, bias=False, device=device, dtype=dtype)
        self.up = nn.Linear(rank, out_features, bias=False, device=device, dtype=dtype)
        self.network_alpha = network_alpha
        self.rank = rank
        self.in_features = in_features
        self.out_features = out_features
        nn.init.normal_(self.down.weight, std=1 / rank)
        nn.init.zeros_(self.up.weight)

    def forward(self, hidden_states, scale=1.0):
        return CustomLoRALinearLayerforward(self, hidden_states, scale)


class LoRAConv2dLayer(nn.Module):
    def __init__(self, in_features, out_features, rank=4, kernel_size=(1, 1), stride=(1, 1), padding=0,
                 network_alpha=None):
        super().__init__()
        self.down = nn.Conv2d(in_features, rank, kernel_size=kernel_size, stride=stride, padding=padding, bias=False)
        self.up = nn.Conv2d(rank, out_features, kernel_size=(1, 1), stride=(1, 1), bias=False)
        self.network_alpha = network_alpha
        self.rank = rank
        nn.init.normal_(self.down.weight, std=1 / rank)
        nn.init.zeros_(self.up.weight)

    def forward(self, hidden_states, scale=1.0):
        return CustomLoRAConv2dLayerforward(self, hidden_states, scale)


class LoRACompatibleLinear(nn.Linear):
    def __init__(self, *args, lora_layer=None, **kwargs):
        super().__init__(*args, **kwargs)
        self.lora_layer = lora_layer

    def set_lora_layer(self, lora_layer):
        self.lora_layer = lora_layer

    def forward(self, hidden_states, scale=1.0, residual=None, geglu=False):
        return CustomLoRACompatibleLinearforward(self, hidden_states, scale, residual=residual, geglu=geglu)


class LoRACompatibleConv(nn.Conv2d):
    def __init__(self, *args, lora_layer=None, **kwargs):
        super().__init__(*args, **kwargs)
        self.lora_layer = lora_layer

    def set_lora_layer(self, lora_layer):
        self.lora_layer = lora_layer

    def forward(self, hidden_states, scale=1.0, residual=None):
        return CustomLoRACompatibleConvforward(self, hidden_states, scale, residual=residual)


# --------------------------------------------------------------------------------------------- LoRA sites
class LoraSite:
    """Compute-side state of one LoRA layer: bf16 A [r,K], A^T, Bup [N,r], Bup^T and fp32 gradient views."""

    def __init__(self, layer):
        self.layer = layer
        self.rank = layer.rank
        self.managed = False  # True when a LoraBank owns storage and refreshes the bf16 copies itself
        self._ver = (-1, -1)
        self.a16 = self.at16 = self.b16 = self.bt16 = None

    def _pd(self):
        return self.layer.down.weight, self.layer.up.weight

    def refresh(self, force=False):
        down, up = self._pd()
        if not down.is_cuda:
            raise ops.L.AqlError("watermark LoRA weights must live on the GPU: the HIP path has no CPU fallback")
        ver = (down._version, up._version)
        if not force and (self.managed or ver == self._ver) and self.a16 is not None:
            return
        r = self.rank
        K = down.numel() // r
        N = up.numel() // r
        if self.a16 is None:
            dev = down.device
            self.a16 = torch.empty(r, K, dtype=torch.bfloat16, device=dev)
            self.at16 = torch.empty(K, r, dtype=torch.bfloat16, device=dev)
            self.b16 = torch.empty(N, r, dtype=torch.bfloat16, device=dev)
            self.bt16 = torch.empty(r, N, dtype=torch.bfloat16, device=dev)
        d32 = down.detach().float().reshape(r, K).contiguous()
        u32 = up.detach().float().reshape(N, r).contiguous()
        L = ops.L
        L.call("aql_cast_transpose", L.ptr(d32), r, K, L.ptr(self.a16), L.ptr(self.at16), L.stream_ptr())
        L.call("aql_cast_transpose", L.ptr(u32), N, r, L.ptr(self.b16), L.ptr(self.bt16), L.stream_ptr())
        self._ver = ver

    @property
    def ga(self):
        down = self.layer.down.weight
        if down.grad is None:
            down.grad = torch.zeros_like(down, dtype=torch.float32)
        return down.grad.view(self.rank, -1)

    @property
    def gb(self):
        up = self.layer.up.weight
        if up.grad is None:
            up.grad = torch.zeros_like(up, dtype=torch.float32)
        return up.grad.view(-1, self.rank)


def _site_of(layer):
    site = getattr(layer, "_aql_site", None)
    if site is None:
        site = LoraSite(layer)
        object.__setattr__(layer, "_aql_site", site)
    site.refresh()
    return site


def _packed_linear(mod):
    pk = getattr(mod, "_aql_packed", None)
    if pk is None:
        pk = ops.PackedLinear(mod.weight, mod.bias)
        object.__setattr__(mod, "_aql_packed", pk)
    return pk


def _packed_conv3(mod):
    pk = getattr(mod, "_aql_packed", None)
    if pk is None:
        pk = ops.PackedConv3x3(mod.weight, mod.bias, mod.stride[0])
        object.__setattr__(mod, "_aql_packed", pk)
    return pk


def _scale16(scale, nb, r, device):
    """-> (bf16 [nb, r] tensor or None, fp32 tensor that carries grad or None)."""
    if scale is None:
        return None
    if isinstance(scale, torch.Tensor):
        if scale.dim() != 2 or scale.shape[1] != r:
            raise ValueError(f"scale must be [batch, rank={r}], got {tuple(scale.shape)}")
        if scale.shape[0] != nb:
            scale = scale.expand(nb, r)
        return scale
    return torch.full((nb, r), float(scale), dtype=torch.bfloat16, device=device)


def _run_linear(x2d, packed, lora_layer, scale, nb, rps, residual, geglu=False):
    site = S = S16 = None
    if lora_layer is not None and scale is not None:
        site = _site_of(lora_layer)
        alpha = getattr(lora_layer, "network_alpha", None)
        if alpha is not None:
            # `up_hidden_states *= network_alpha / rank` (lora_modules.py:21-22, 39-40) commutes with the diagonal scale: fold
            # it into S (autograd carries the factor back to dS).  Always None on the PPFT path (ppft_train.py:662-666).
            k = float(alpha) / site.rank
            scale = scale * k if torch.is_tensor(scale) else float(scale) * k
        S = _scale16(scale, nb, site.rank, x2d.device)
        S16 = getattr(S, "_aql_s16", None)  # bf16 copy made once per scale tensor, shared by all 192 sites
        if S16 is None:
            S16 = S.detach().to(torch.bfloat16).contiguous()
            S._aql_s16 = S16
    return ops.lora_linear(x2d, packed, site, S, S16, rps, residual, geglu)


# ------------------------------------------------------------------------------------ the four forwards
def CustomLoRACompatibleLinearforward(self, hidden_states, scale=1.0, residual=None, geglu=False):
    """nn.Linear.forward(x) [+ lora_layer(x, scale)]  -- one fused MFMA GEMM (reference lora_modules.py:56-62).
    ``geglu=True`` (the host is GEGLU.proj): returns  y[..., :F] * gelu(y[..., F:])  with the activation applied in the
    GEMM epilogue (scripts/lib/original_unet.py:727-729)."""
    shp = hidden_states.shape
    x2d = hidden_states.reshape(-1, shp[-1])
    if x2d.dtype != torch.bfloat16:
        x2d = x2d.to(torch.bfloat16)
    x2d = x2d.contiguous()
    nb = shp[0] if hidden_states.dim() >= 2 else 1
    rps = x2d.shape[0] // nb
    res2d = None if residual is None else residual.reshape(-1, self.out_features).contiguous()
    y = _run_linear(x2d, _packed_linear(self), self.lora_layer, scale, nb, rps, res2d, geglu)
    return y.reshape(*shp[:-1], y.shape[-1])


def CustomLoRACompatibleConvforward(self, hidden_states, scale=1.0, residual=None):
    """Conv2d [+ lora_layer(x, scale)] (reference lora_modules.py:46-54).  1x1 convs run as the fused LoRA GEMM on
    the channels-last token view; 3x3/pad-1 convs (never LoRA'd in unet_keys.json) run the implicit-GEMM kernel."""
    k = self.kernel_size
    if k == (1, 1) and self.stride == (1, 1) and self.padding in ((0, 0), 0):
        x = ops.as_cl(hidden_states if hidden_states.dtype == torch.bfloat16 else hidden_states.to(torch.bfloat16))
        B, C, H, W = x.shape
        x2d = ops.nhwc_view(x).reshape(B * H * W, C)
        res2d = None if residual is None else ops.nhwc_view(ops.as_cl(residual)).reshape(B * H * W, self.out_channels)
        y = _run_linear(x2d, _packed_linear(self), self.lora_layer, scale, B, H * W, res2d)
        return y.view(B, H, W, self.out_channels).permute(0, 3, 1, 2)
    if k == (3, 3) and self.padding in ((1, 1), 1) and self.stride in ((1, 1), (2, 2)) and self.lora_layer is None:
        x = hidden_states if hidden_states.dtype == torch.bfloat16 else hidden_states.to(torch.bfloat16)
        return ops.conv3x3(x, _packed_conv3(self), False, None, residual)
    raise NotImplementedError(f"conv geometry k={k} stride={self.stride} pad={self.padding} "
                              f"lora={self.lora_layer is not None} is not on the PPFT path")


def CustomLoRALinearLayerforward(self, hidden_states, scale=1.0):
    """up(down(x) @ diag_embed(scale)) -- the LoRA branch alone (reference lora_modules.py:9-26)."""
    shp = hidden_states.shape
    orig = hidden_states.dtype
    x2d = hidden_states.reshape(-1, shp[-1]).to(torch.bfloat16).contiguous()
    nb = shp[0]
    pk = getattr(self, "_aql_zero_base", None)
    if pk is None:
        N, K = self.up.weight.shape[0], self.down.weight.numel() // self.rank
        pk = ops.PackedLinear(torch.zeros(N, K, device=x2d.device), None)
        object.__setattr__(self, "_aql_zero_base", pk)
    y = _run_linear(x2d, pk, self, scale, nb, x2d.shape[0] // nb, None)
    return y.reshape(*shp[:-1], y.shape[-1]).to(orig)


def CustomLoRAConv2dLayerforward(self, hidden_states, scale=1.0):
    """up(down(x) * scale[:, :, None, None]) for 1x1 convs (reference lora_modules.py:28-44)."""
    orig = hidden_states.dtype
    x = ops.as_cl(hidden_states.to(torch.bfloat16))
    B, C, H, W = x.shape
    y = CustomLoRALinearLayerforward(self, ops.nhwc_view(x).reshape(B, H * W, C), scale)
    return y.view(B, H, W, -1).permute(0, 3, 1, 2).to(orig)


# --------------------------------------------------------------------------- injection / patching loops
def load_unet_keys(unet):
    """The reference reads utils/unet_keys.json (192 sorted module paths); here the same list is derived from the
    module tree (tests/test_golden.py checks it equals the reference file)."""
    from .unet import lora_keys
    return lora_keys(unet)


def _walk(root, key):
    m = root
    for sub in key.split("."):
        m = getattr(m, sub)
    return m


def inject_lora(unet, rank, keys=None, lora_state=None):
    """ppft_train.py:620-678: build a LoRA layer for each key, optionally load weights, set_lora_layer.
    Returns the list of trainable parameters (2 per site, in key order)."""
    keys = keys if keys is not None else load_unet_keys(unet)
    params = []
    dev = next(unet.parameters()).device
    for key in keys:
        host = _walk(unet, key)
        if isinstance(host, LoRACompatibleConv):
            lora = LoRAConv2dLayer(host.in_channels, host.out_channels, rank=rank, kernel_size=host.kernel_size,
                                   stride=host.stride, padding=host.padding)
        elif isinstance(host, LoRACompatibleLinear):
            lora = LoRALinearLayer(host.in_features, host.out_features, rank)
        else:
            raise ValueError(f"Module {key} is not a LoRACompatibleConv or LoRACompatibleLinear module.")
        if lora_state is not None:
            lora.load_state_dict({"down.weight": lora_state[key + ".down.weight"],
                                  "up.weight": lora_state[key + ".up.weight"]})
        lora.to(device=dev, dtype=torch.float32)
        host.set_lora_layer(lora)
        params.extend(lora.parameters())
    return params


def patch_lora_forwards(unet):
    """ppft_train.py:681-689: bind the custom forwards onto every LoRA-compatible module of the U-Net."""
    for _, module in unet.named_modules():
        if isinstance(module, LoRACompatibleConv):
            module.forward = types.MethodType(CustomLoRACompatibleConvforward, module)
            if module.lora_layer is not None:
                module.lora_layer.forward = types.MethodType(CustomLoRAConv2dLayerforward, module.lora_layer)
        elif isinstance(module, LoRACompatibleLinear):
            module.forward = types.MethodType(CustomLoRACompatibleLinearforward, module)
            if module.lora_layer is not None:
                module.lora_layer.forward = types.MethodType(CustomLoRALinearLayerforward, module.lora_layer)


# ------------------------------------------------------------------------------------------ flat bank
def _is_kv(k):
    return k.endswith(".attn2.to_k") or k.endswith(".attn2.to_v")


def backward_stage(key):
    """Which leg of backward completes a site's weight gradients: 0 = up path, 1 = mid block + down_blocks.3 / .2, 2 = down_blocks.1,
    3 = down_blocks.0.  The U-Net fires a backward hook at the end of legs 0, 1 and 2 (unet.forward: gradients of the mid-block
    output, of down_blocks.1's output, of down_blocks.0's output)."""
    if key.startswith("up_blocks."):
        return 0
    if key.startswith("mid_block."):
        return 1
    if key.startswith("down_blocks."):
        i = int(key.split(".")[1])
        return 1 if i >= 2 else (2 if i == 1 else 3)
    return 3


N_STAGES = 4


def bank_stages(keys, kv_last=True):
    """-> (site indices in buffer order, [number of leading sites complete after backward leg 0, 1, 2]).  Pure host logic
    (CPU-tested): reverse traversal order; ``kv_last`` moves the text-state projections attn2.to_k / to_v to the end of the buffer
    (rank 32: all 32 of them are ONE grouped launch in front of the U-Net whose backward is the last node of the graph; at other
    ranks they are ordinary sites of their block and stay in traversal order)."""
    rev = list(reversed(range(len(keys))))
    if kv_last:
        order = [i for i in rev if not _is_kv(keys[i])] + [i for i in rev if _is_kv(keys[i])]
    else:
        order = rev
    ends, pos = [], 0
    for stage in range(N_STAGES - 1):
        while pos < len(order) and backward_stage(keys[order[pos]]) <= stage and not (kv_last and _is_kv(keys[order[pos]])):
            pos += 1
        ends.append(pos)
    return order, ends


def bank_order(keys):
    """-> (site indices in buffer order, number of leading sites that belong to the up path): `bank_stages` with the text-state
    projections last, first cut only."""
    order, ends = bank_stages(keys, True)
    return order, ends[0]


class LoraBank:
    """Re-homes every LoRA parameter of a U-Net into ONE flat fp32 buffer (+ flat grad / exp_avg / exp_avg_sq),
    so that clip-norm, AdamW and the data-parallel gradient exchange are single kernels / collectives over
    contiguous HBM instead of 384 small tensors.  Sites are laid out in REVERSE key order so that the gradients
    that finish first in backward (up_blocks.3 ...) sit at the front of the buffer: bucket i of the exchange is
    complete as soon as backward has passed its last site."""

    def __init__(self, unet, keys=None, extra_params=(), kv_last=None):
        keys = keys if keys is not None else load_unet_keys(unet)
        self.layers = [_walk(unet, k).lora_layer for k in keys]
        if kv_last is None:    # the text-state projections are grouped in front of the U-Net (unet._ctx_kv) at rank 32 only
            kv_last = all(int(l.rank) == 32 for l in self.layers)
        # Gradient-ready order: reverse traversal, EXCEPT the text-state projections attn2.to_k / to_v -- at rank 32 all 32 of
        # them are one grouped launch in front of the U-Net (unet._ctx_kv) whose backward is the LAST node of the graph, so
        # their gradients complete last wherever their block sits: they go to the end of the buffer.  The buffer then reads
        #   [ up_blocks sites | mid + down_blocks sites | text k|v sites | extra (mapper) ]
        # and the first region (`n_early` elements) is complete the moment backward leaves the up path: the overlapped
        # exchange all-reduces it under the mid / down backward (ppft.PPFTTrainer, DDP's early buckets at ppft_train.py:1058).
        order, ends = bank_stages(keys, kv_last)
        plist = []
        self.cuts = [0] * len(ends)     # elements complete after backward leg 0, 1, 2 (lora.backward_stage)
        for j, i in enumerate(order):
            plist += [self.layers[i].down.weight, self.layers[i].up.weight]
            for q, e in enumerate(ends):
                if j < e:
                    self.cuts[q] += self.layers[i].down.weight.numel() + self.layers[i].up.weight.numel()
        self.n_early = self.cuts[0]
        self.n_lora = sum(p.numel() for p in plist)
        plist += list(extra_params)
        self.params = plist
        n = sum(p.numel() for p in plist)
        dev = plist[0].device
        pad = (-n) % 64
        self.flat = torch.zeros(n + pad, dtype=torch.float32, device=dev)
        self.grad = torch.zeros_like(self.flat)
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.numel = n
        off = 0
        self.offsets = []
        for p in plist:
            k = p.numel()
            self.flat[off:off + k].copy_(p.detach().float().reshape(-1))
            p.data = self.flat[off:off + k].view(p.shape)
            p.grad = self.grad[off:off + k].view(p.shape)
            self.offsets.append(off)
            off += k
        # bf16 compute copies (A, A^T, Bup, Bup^T per site) in one flat buffer, refreshed by ONE batched launch
        import numpy as np
        total16 = sum(2 * (l.down.weight.numel() + l.up.weight.numel()) for l in self.layers)
        self.c16 = torch.empty(total16 + 64 + 8 * 4 * len(self.layers), dtype=torch.bfloat16, device=dev)
        desc = np.zeros(2 * len(self.layers), dtype=np.dtype(
            [("w", "<u8"), ("out", "<u8"), ("outT", "<u8"), ("rows", "<i4"), ("cols", "<i4"), ("first", "<i4"),
             ("pad", "<i4")]))
        self.sites = []
        off16, tile, di = 0, 0, 0

        def take(n):
            nonlocal off16
            v = self.c16[off16:off16 + n]
            off16 += (n + 7) // 8 * 8  # keep every view 16-byte aligned
            return v

        # Cohorts: sites whose linears read the same input run as ONE grouped launch (ops.GroupedLoraFn) and need their bf16
        # A matrices stacked back to back (and their Bup matrices likewise): q|k|v of every self-attention, and the k|v
        # projections of the text states of ALL cross-attentions (sorted key order = U-Net traversal order, k before v).
        cohort_of, cohorts = {}, []
        by_key = dict(zip(keys, self.layers))
        for k in keys:
            if k.endswith(".attn1.to_q"):
                base = k[:-len("to_q")]
                trio = [base + "to_q", base + "to_k", base + "to_v"]
                if all(t in by_key for t in trio):
                    cohorts.append(trio)
        kv = sorted([k for k in keys if k.endswith(".attn2.to_k") or k.endswith(".attn2.to_v")],
                    key=lambda k: (k.rsplit(".", 1)[0], k.rsplit(".", 1)[1]))
        if kv:
            cohorts.append(kv)
        for c in cohorts:
            for k in c:
                cohort_of[k] = c
        self.cohorts = cohorts
        views, done = {}, set()
        for key, layer in zip(keys, self.layers):
            if key in done:
                continue
            members = cohort_of.get(key, [key])
            dims = {}
            for m in members:
                lay = by_key[m]
                r = lay.rank
                dims[m] = (r, lay.down.weight.numel() // r, lay.up.weight.numel() // r)
            a = {m: take(dims[m][0] * dims[m][1]).view(dims[m][0], dims[m][1]) for m in members}     # stacked A [r, K]
            b = {m: take(dims[m][2] * dims[m][0]).view(dims[m][2], dims[m][0]) for m in members}     # stacked Bup [N, r]
            r0, K0, N0 = dims[members[0]]
            wide_trio = len(members) == 3 and not _is_kv(members[0]) and r0 > 32 and all(dims[m] == (r0, K0, N0) for m in members)
            if wide_trio:
                # q | k | v at a rank above 32 (ops.GroupedWideFn): the transposed copies are laid out for the grouped backward too --
                # A^T of the three sites as COLUMN blocks of one [K, 3r] matrix (the B operand of dX += [dT_q | dT_k | dT_v].[A_q; A_k; A_v]),
                # Bup^T stacked [3r, N] (the B operand of [dTs_q | dTs_k | dTs_v] = dY_g.Bup_g)
                atc = take(K0 * 3 * r0).view(K0, 3 * r0)
                btc = take(3 * r0 * N0).view(3 * r0, N0)
            wide_kv = len(members) >= 2 and _is_kv(members[0]) and r0 > 32   # the text k | v cohort: Bup^T of all members back to back (k, v of one attention adjacent)
            bt = {m: take(dims[m][2] * dims[m][0]).view(dims[m][0], dims[m][2]) for m in members} if wide_kv else {}
            for g_, m in enumerate(members):
                r, K, N = dims[m]
                if wide_trio:
                    views[m] = (a[m], atc[:, g_ * r:(g_ + 1) * r], b[m], btc[g_ * r:(g_ + 1) * r])
                elif wide_kv:
                    views[m] = (a[m], take(r * K).view(K, r), b[m], bt[m])
                else:
                    views[m] = (a[m], take(r * K).view(K, r), b[m], take(N * r).view(r, N))
                done.add(m)
        for key, layer in zip(keys, self.layers):
            site = LoraSite(layer)
            site.managed = True
            r = layer.rank
            K = layer.down.weight.numel() // r
            N = layer.up.weight.numel() // r
            site.a16, site.at16, site.b16, site.bt16 = views[key]
            for w, rows, cols, o, ot in ((layer.down.weight, r, K, site.a16, site.at16),
                                         (layer.up.weight, N, r, site.b16, site.bt16)):
                # (last field: leading dimension of the transposed copy when it is a column block of a wider matrix, 0 = dense)
                desc[di] = (w.data_ptr(), o.data_ptr(), ot.data_ptr(), rows, cols, tile, 0 if ot.stride(0) == rows else ot.stride(0))
                tile += ((rows + 31) // 32) * ((cols + 31) // 32)
                di += 1
            object.__setattr__(layer, "_aql_site", site)
            self.sites.append(site)
        assert off16 <= self.c16.numel()
        self._desc = torch.from_numpy(desc.view(np.uint8).copy()).to(dev)
        self._ndesc, self._ntiles = di, tile
        self.refresh()

    def refresh(self):
        """Re-cast the fp32 masters to the bf16 compute copies (call after every optimizer step): one launch."""
        L = ops.L
        L.call("aql_cast_transpose_batched", L.ptr(self._desc), self._ndesc, self._ntiles, L.stream_ptr())

    def zero_grad(self):
        self.grad.zero_()
