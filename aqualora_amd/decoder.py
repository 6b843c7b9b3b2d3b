"""SecretDecoder (reference utils/models.py:84-96; duplicate evaluation/utils_eval.py:142-154): torchvision
``efficientnet_b1`` with ``classifier[1] = Linear(1280, 2*bits)``; ``forward(x)`` resizes to 512x512 (bilinear) and
returns logits ``[B, bits, 2]``; callers take ``argmax(-1)`` as the message bits.

torchvision is not installed on the target image, so the module tree below restates torchvision 0.15's public
EfficientNet-B1 layout (stem 3->32 s2; MBConv stages (t,k,s,c,n) = (1,3,1,16,2) (6,3,2,24,3) (6,5,2,40,3) (6,3,2,80,4)
(6,5,1,112,4) (6,5,2,192,5) (6,3,1,320,2); SE squeeze = block-input/4; head 320->1280; BN eps 1e-5) with the same
state-dict key names as ``msgdecoder.pt`` (``model.features.*``, ``model.classifier.1.*``).  PARITY UNPINNED against
torchvision itself: HIP is checked against oracle/decoder_oracle.py, a CPU restatement of the same
public architecture.

Inference (eval mode: BN running stats, no dropout / stochastic depth) runs in the fused fp32 HIP kernels of
csrc/aql_decoder.hip with BatchNorm folded into the convolutions.  Training mode (``.train()``: stage 1,
latent_wm_pretrain.py:159-225, and the robustness fine-tune) runs layer by layer through csrc/aql_decoder_train.hip --
BatchNorm with batch statistics and running-stat updates, stochastic depth (torchvision "row" mode, p = 0.2*i/23),
dropout 0.2 -- with every backward in HIP; torch only adds the residual branches and owns the parameters.
"""
import os

import torch
import torch.nn as nn

from . import _lib as L

B1_STAGES = [(1, 3, 1, 32, 16, 2), (6, 3, 2, 16, 24, 3), (6, 5, 2, 24, 40, 3), (6, 3, 2, 40, 80, 4),
             (6, 5, 1, 80, 112, 4), (6, 5, 2, 112, 192, 5), (6, 3, 1, 192, 320, 2)]  # t, k, s, cin, cout, n


def _cna(cin, cout, k, stride, groups=1):
    """Conv2dNormActivation: index 0 conv (no bias), 1 BatchNorm2d (activation has no parameters)."""
    return nn.Sequential(nn.Conv2d(cin, cout, k, stride, (k - 1) // 2, groups=groups, bias=False),
                         nn.BatchNorm2d(cout, eps=1e-5))


class _SE(nn.Module):
    def __init__(self, c, cs):
        super().__init__()
        self.fc1 = nn.Conv2d(c, cs, 1)
        self.fc2 = nn.Conv2d(cs, c, 1)


class _MBConv(nn.Module):
    def __init__(self, t, k, s, cin, cout):
        super().__init__()
        cexp = cin * t
        layers = []
        if t != 1:
            layers.append(_cna(cin, cexp, 1, 1))
        layers.append(_cna(cexp, cexp, k, s, groups=cexp))
        layers.append(_SE(cexp, max(1, cin // 4)))
        layers.append(_cna(cexp, cout, 1, 1))
        self.block = nn.Sequential(*layers)
        self.use_res = (s == 1 and cin == cout)
        self.cfg = (t, k, s, cin, cout)


class _EfficientNetB1(nn.Module):
    def __init__(self, num_out):
        super().__init__()
        feats = [_cna(3, 32, 3, 2)]
        for (t, k, s, cin, cout, n) in B1_STAGES:
            feats.append(nn.Sequential(*[_MBConv(t, k, s if i == 0 else 1, cin if i == 0 else cout, cout)
                                         for i in range(n)]))
        feats.append(_cna(320, 1280, 1, 1))
        self.features = nn.Sequential(*feats)
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.classifier = nn.Sequential(nn.Dropout(0.2), nn.Linear(1280, num_out))


def train_step_algorithmic_bytes(B, res=512, fused=False):
    """HBM bytes of one SecretDecoder training step (forward + backward-data + backward-weight) on B images of res x res, fp32, counted
    OP BY OP with no fusion between ops: forward = every conv / BatchNorm+SiLU / squeeze-excite scale / residual add reads its input(s)
    and writes its output once; backward-data = (dy, saved x) -> dx per op; backward-weight of the convolutions = (dy, x) again.
    Weights, the squeeze-excite vectors and the classifier are negligible beside the maps.  This is the roofline the step as built can
    reach (bench.py config5.roofline); fusing BatchNorm into its producer / consumer would lower the floor itself.
    ``fused=True``: the floor of a step whose row-local ops ride in their neighbours (round 6) -- train-mode
    BatchNorm still needs its batch statistics before anything can be normalised, so per BatchNorm one statistics read remains forward
    (the producing conv's epilogue could carry even that) and one (dy, x) pass backward; BatchNorm-apply + SiLU is recomputed on the fly
    by the consumer's loader (no normalised map is written or read), the squeeze-excite gate multiplies in the project conv's loader,
    the residual add sits in its epilogue."""
    total = [0]

    def op(n_in, n_out, conv=False, kind=None):
        if fused and not conv:
            if kind == "bn":         # statistics pass forward; (dy, x) -> sums backward; the apply passes ride in the neighbours
                total[0] += 4 * (n_in + 2 * n_in)
            elif kind == "pool":     # squeeze: one read forward, its backward rides in the excite backward
                total[0] += 4 * n_in
            # excite / residual add: in the project conv's loader / epilogue (the residual's read is counted there)
            elif kind == "res":
                total[0] += 4 * (n_out + n_out)      # forward: the skip tensor read in the epilogue; backward: dy fans out (one extra read)
            return
        fwd = n_in + n_out
        bwd = n_out + n_in + n_in
        total[0] += 4 * (fwd + bwd + ((n_out + n_in) if conv else 0))

    H = res // 2
    op(B * res * res * 3, B * H * H * 32, True)      # stem conv
    op(B * H * H * 32, B * H * H * 32, kind="bn")    # its BatchNorm + SiLU
    for (t, k, s, cin, cout, n) in B1_STAGES:
        for i in range(n):
            st, ci = (s if i == 0 else 1), (cin if i == 0 else cout)
            ce, Ho = ci * t, H // st
            if t != 1:
                op(B * H * H * ci, B * H * H * ce, True)
                op(B * H * H * ce, B * H * H * ce, kind="bn")
            op(B * H * H * ce, B * Ho * Ho * ce, True)   # depthwise
            op(B * Ho * Ho * ce, B * Ho * Ho * ce, kind="bn")       # BatchNorm + SiLU
            op(B * Ho * Ho * ce, B * ce, kind="pool")                 # squeeze (pool)
            op(B * Ho * Ho * ce, B * Ho * Ho * ce, kind="excite")       # excite (scale by the gate)
            op(B * Ho * Ho * ce, B * Ho * Ho * cout, True)
            op(B * Ho * Ho * cout, B * Ho * Ho * cout, kind="bn")   # BatchNorm
            if st == 1 and ci == cout:
                op(2 * B * Ho * Ho * cout, B * Ho * Ho * cout, kind="res")   # residual add (stochastic depth)
            H = Ho
    op(B * H * H * 320, B * H * H * 1280, True)
    op(B * H * H * 1280, B * H * H * 1280, kind="bn")
    op(B * H * H * 1280, B * 1280, kind="pool")
    return total[0]


def fold_bn(conv_w, bn):
    """eval-mode BatchNorm folded into the preceding bias-free conv: w' = w*g/sqrt(var+eps), b' = beta - mean*g/..."""
    g = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
    w = conv_w.detach().float() * g.view(-1, *([1] * (conv_w.dim() - 1)))
    b = bn.bias.detach().float() - bn.running_mean.detach().float() * g
    return w, b


class SecretDecoder(nn.Module):
    def __init__(self, output_size=64):
        super().__init__()
        self.output_size = output_size
        self.model = _EfficientNetB1(output_size * 2)
        self._packed = None

    # ---------------------------------------------------------------------------------------- packing
    def _pack(self):
        m = self.model
        dev = m.classifier[1].weight.device
        P = {}
        w, b = fold_bn(m.features[0][0].weight, m.features[0][1])
        P["stem"] = (w.permute(2, 3, 1, 0).reshape(27, -1).contiguous(), b.contiguous())  # [(kh,kw,ci)][co]
        blocks = []
        for stage in list(m.features)[1:-1]:
            for blk in stage:
                t, k, s, cin, cout = blk.cfg
                lay = list(blk.block)
                d = {"cfg": blk.cfg, "res": blk.use_res}
                i = 0
                if t != 1:
                    w, b = fold_bn(lay[0][0].weight, lay[0][1])
                    d["exp"] = (w.reshape(w.shape[0], -1).contiguous(), b.contiguous())
                    i = 1
                w, b = fold_bn(lay[i][0].weight, lay[i][1])
                d["dw"] = (w.reshape(w.shape[0], -1).t().contiguous(), b.contiguous())  # [k*k][C]
                se = lay[i + 1]
                d["se"] = (se.fc1.weight.detach().float().reshape(se.fc1.weight.shape[0], -1).contiguous(),
                           se.fc1.bias.detach().float().contiguous(),
                           se.fc2.weight.detach().float().reshape(se.fc2.weight.shape[0], -1).contiguous(),
                           se.fc2.bias.detach().float().contiguous())
                w, b = fold_bn(lay[i + 2][0].weight, lay[i + 2][1])
                d["proj"] = (w.reshape(w.shape[0], -1).contiguous(), b.contiguous())
                blocks.append(d)
        P["blocks"] = blocks
        w, b = fold_bn(m.features[-1][0].weight, m.features[-1][1])
        P["head"] = (w.reshape(w.shape[0], -1).contiguous(), b.contiguous())
        P["fc"] = (m.classifier[1].weight.detach().float().contiguous(), m.classifier[1].bias.detach().float().contiguous())
        P["device"] = dev
        self._packed = P
        return P

    def train(self, mode=True):
        self._packed = None  # folded inference weights go stale as soon as the parameters may move
        return super().train(mode)

    def __getstate__(self):
        """copy.deepcopy / pickle: the folded weights and their captured graphs are derived state (a HIP graph cannot be copied)."""
        state = dict(self.__dict__)
        state["_packed"] = None
        return state

    # ---------------------------------------------------------------------------------------- forward
    def forward(self, x, sd_noise=None, drop_mask=None):
        if not x.is_cuda:
            raise L.AqlError("SecretDecoder: the HIP path needs GPU tensors; there is no CPU fallback")
        if self.training:
            return self.forward_train(x, sd_noise, drop_mask)
        with torch.no_grad():
            return self._forward_eval(x)

    def forward_train(self, x, sd_noise=None, drop_mask=None):
        """train()-mode forward with autograd.  ``sd_noise`` ([n_blocks][B] survival factors, already divided by the
        survival probability) and ``drop_mask`` ([B,1280], already divided by 0.8) override the random draws (tests)."""
        m = self.model
        B = x.shape[0]
        h = _ResizeFn.apply(x.float().contiguous(), 512, 512)            # [B,512,512,3] NHWC
        h = _StemFn.apply(h, m.features[0][0].weight)                     # [B,256,256,32]
        h = _bn_act(h, m.features[0][1], True)
        nblk = sum(len(stage) for stage in list(m.features)[1:-1])
        bi = 0
        for stage in list(m.features)[1:-1]:
            for blk in stage:
                t, k, s, cin, cout = blk.cfg
                lay = list(blk.block)
                inp = h
                i = 0
                if t != 1:
                    h = _conv1x1(h, lay[0][0].weight, None)
                    h = _bn_act(h, lay[0][1], True)
                    i = 1
                h = _DwConvFn.apply(h, lay[i][0].weight, k, s)
                h = _bn_act(h, lay[i][1], True)
                se = lay[i + 1]
                Bq, Hc, Wc, C = h.shape
                pooled = _ChanReduceFn.apply(h.view(Bq, Hc * Wc, C))      # [B,C] mean
                g = _conv1x1(pooled, se.fc1.weight, se.fc1.bias)
                g = _ActFn.apply(g, 1)
                g = _conv1x1(g, se.fc2.weight, se.fc2.bias)
                g = _ActFn.apply(g, 2)
                h = _ChanScaleFn.apply(h.view(Bq, Hc * Wc, C), g).view(Bq, Hc, Wc, C)
                h = _conv1x1(h, lay[i + 2][0].weight, None)
                if blk.use_res:
                    p = 0.2 * bi / nblk                                   # torchvision: sd_prob * block_id / total
                    if sd_noise is not None:
                        noise = sd_noise[bi].to(h.device, torch.float32)
                    else:
                        noise = torch.empty(B, device=h.device).bernoulli_(1.0 - p)
                        if p < 1.0:
                            noise = noise / (1.0 - p)
                    if FUSE_BN_RES:     # BatchNorm + stochastic depth + skip in the BatchNorm's apply pass (round 6)
                        h = _bn_res(h, lay[i + 2][1], inp, noise)
                    else:
                        h = _bn_act(h, lay[i + 2][1], False)
                        Bq, Hc, Wc, C = h.shape
                        h = _ChanScaleFn.apply(h.view(Bq, Hc * Wc, C), noise.view(B, 1).expand(B, C).contiguous())
                        h = h.view(Bq, Hc, Wc, C) + inp
                else:
                    h = _bn_act(h, lay[i + 2][1], False)
                bi += 1
        h = _conv1x1(h, m.features[-1][0].weight, None)
        h = _bn_act(h, m.features[-1][1], True)
        Bq, Hc, Wc, C = h.shape
        pooled = _ChanReduceFn.apply(h.view(Bq, Hc * Wc, C))
        if drop_mask is None:
            drop_mask = torch.empty(B, C, device=h.device).bernoulli_(0.8) / 0.8
        pooled = _ChanScaleFn.apply(pooled.view(B, 1, C), drop_mask.to(h.device, torch.float32).contiguous()).view(B, C)
        logits = _conv1x1(pooled, m.classifier[1].weight, m.classifier[1].bias)
        return logits.view(-1, self.output_size, 2)

    def _forward_eval(self, x):
        """Inference (utils_eval.py:131-140): the ~125 launches of the folded network as ONE HIP graph per input shape, kept with the
        packed weights (so `.train()` or a re-pack drops it).  A single image is otherwise bound by the host's launch rate (5 ms of
        Python for 1.5 ms of kernels).  AQL_DECODER_GRAPH=0 or an ongoing stream capture runs the launches directly."""
        P = self._packed or self._pack()
        if os.environ.get("AQL_DECODER_GRAPH", "1") == "0" or torch.cuda.is_current_stream_capturing():
            return self._eval_launches(P, x)
        key = (tuple(x.shape), x.dtype, x.device.index)
        graphs = P.setdefault("graphs", {})
        ent = graphs.get(key)
        if ent is None:
            xin = torch.empty(x.shape, dtype=x.dtype, device=x.device)
            xin.copy_(x)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self._eval_launches(P, xin)                  # warm-up outside the capture
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                out = self._eval_launches(P, xin)
            while len(graphs) >= 3:
                graphs.pop(next(iter(graphs)))
            ent = graphs[key] = (g, xin, out)
        g, xin, out = ent
        xin.copy_(x)
        g.replay()
        return out.clone()

    def _eval_launches(self, P, x):
        st = L.stream_ptr()
        B, C, H, W = x.shape
        dev = x.device
        x = x.float().contiguous()
        cur = torch.empty(B, 512, 512, 3, device=dev)
        L.call("aql_resize_bilinear_nhwc", L.ptr(x), B, 3, H, W, 512, 512, L.ptr(cur), st)
        h = torch.empty(B, 256, 256, 32, device=dev)
        L.call("aql_stem_conv3x3s2_silu", L.ptr(cur), L.ptr(P["stem"][0]), L.ptr(P["stem"][1]), B, 512, 512, 32,
               L.ptr(h), st)
        Hc = Wc = 256
        for d in P["blocks"]:
            t, k, s, cin, cout = d["cfg"]
            cexp = cin * t
            inp = h
            if t != 1:
                e = torch.empty(B, Hc, Wc, cexp, device=dev)
                L.call("aql_pwconv_f32", L.ptr(h), L.ptr(d["exp"][0]), L.ptr(d["exp"][1]), None, 0, None,
                       B * Hc * Wc, cexp, cin, 1, L.ptr(e), st)
                h = e
            Ho = (Hc + 2 * (k // 2) - k) // s + 1
            dw = torch.empty(B, Ho, Ho, cexp, device=dev)
            L.call("aql_dwconv_silu", L.ptr(h), L.ptr(d["dw"][0]), L.ptr(d["dw"][1]), B, Hc, Wc, cexp, k, s, L.ptr(dw), st)
            Hc = Wc = Ho
            # pool in pixel slabs (a chip-wide pass at any batch size), summed in slab order inside the gate kernel
            S = max(1, min(128, (Hc * Wc) // 64, 2048 // (B * ((cexp + 63) // 64)) or 1))
            part = torch.empty(B, S, cexp, device=dev)
            L.call("aql_avgpool_nhwc_slabs", L.ptr(dw), B, Hc * Wc, cexp, S, L.ptr(part), st)
            gate = torch.empty(B, cexp, device=dev)
            w1, b1, w2, b2 = d["se"]
            L.call("aql_se_gate_slabs", L.ptr(part), S, Hc * Wc, L.ptr(w1), L.ptr(b1), L.ptr(w2), L.ptr(b2), B, cexp, w1.shape[0],
                   L.ptr(gate), st)
            out = torch.empty(B, Hc, Wc, cout, device=dev)
            L.call("aql_pwconv_f32", L.ptr(dw), L.ptr(d["proj"][0]), L.ptr(d["proj"][1]), L.ptr(gate), Hc * Wc,
                   L.ptr(inp) if d["res"] else None, B * Hc * Wc, cout, cexp, 0, L.ptr(out), st)
            h = out
        hd = torch.empty(B, Hc, Wc, 1280, device=dev)
        L.call("aql_pwconv_f32", L.ptr(h), L.ptr(P["head"][0]), L.ptr(P["head"][1]), None, 0, None, B * Hc * Wc, 1280,
               320, 1, L.ptr(hd), st)
        pool = torch.empty(B, 1280, device=dev)
        L.call("aql_avgpool_nhwc", L.ptr(hd), B, Hc * Wc, 1280, L.ptr(pool), st)
        logits = torch.empty(B, self.output_size * 2, device=dev)
        L.call("aql_pwconv_f32", L.ptr(pool), L.ptr(P["fc"][0]), L.ptr(P["fc"][1]), None, 0, None, B,
               self.output_size * 2, 1280, 0, L.ptr(logits), st)
        return logits.view(-1, self.output_size, 2)


# ------------------------------------------------------------------------------------------------------------
# train()-mode building blocks: one autograd.Function per HIP layer (csrc/aql_decoder_train.hip)
# ------------------------------------------------------------------------------------------------------------
def _gemm(A, sam, sak, Bm, sbn, sbk, bias, M, N, K, out):
    L.call("aql_gemm_f32", L.ptr(A), sam, sak, L.ptr(Bm), sbn, sbk, L.ptr(bias), L.ptr(out), N, M, N, K, L.stream_ptr())
    return out


class _Conv1x1Fn(torch.autograd.Function):
    """y[..., Co] = x[..., Ci] . W[Co,Ci]^T (+ bias): 1x1 convolution on channels-last activations / nn.Linear."""

    @staticmethod
    def forward(ctx, x, w, bias):
        x = x.contiguous()
        w2 = w.detach().reshape(w.shape[0], -1).float().contiguous()
        Ci, Co = w2.shape[1], w2.shape[0]
        M = x.numel() // Ci
        y = torch.empty(*x.shape[:-1], Co, device=x.device, dtype=torch.float32)
        _gemm(x, Ci, 1, w2, Ci, 1, None if bias is None else bias.detach().float().contiguous(), M, Co, Ci, y)
        ctx.save_for_backward(x, w2)
        ctx.has_bias = bias is not None
        ctx.wshape = w.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w2 = ctx.saved_tensors
        dy = dy.contiguous()
        Co, Ci = w2.shape
        M = x.numel() // Ci
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = _gemm(dy, Co, 1, w2, 1, Ci, None, M, Ci, Co, torch.empty_like(x))
        if ctx.needs_input_grad[1]:
            dw = _gemm(dy, 1, Co, x, 1, Ci, None, Co, Ci, M, torch.empty(Co, Ci, device=x.device)).view(ctx.wshape)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = torch.empty(Co, device=x.device)
            L.call("aql_chan_reduce", L.ptr(dy), None, 1, M, Co, 1.0, L.ptr(db), L.stream_ptr())
        return dx, dw, db


def _conv1x1(x, w, bias):
    return _Conv1x1Fn.apply(x, w, bias)


def _bn_scratch(M, C, dev):
    return torch.empty(2 * ((M + 2047) // 2048) * C, device=dev, dtype=torch.float32)


class _BNActFn(torch.autograd.Function):
    """nn.BatchNorm2d in training mode (batch statistics, running-stat update) + optional SiLU, on [.., C]."""

    @staticmethod
    def forward(ctx, x, gamma, beta, run_mean, run_var, eps, momentum, act):
        x = x.contiguous()
        C = x.shape[-1]
        M = x.numel() // C
        y = torch.empty_like(x)
        mean = torch.empty(C, device=x.device)
        invstd = torch.empty(C, device=x.device)
        g, b = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        L.call("aql_bn_train_fwd", L.ptr(x), L.ptr(g), L.ptr(b), M, C, float(eps), float(momentum), int(act), L.ptr(y),
               L.ptr(mean), L.ptr(invstd), L.ptr(run_mean), L.ptr(run_var), L.ptr(_bn_scratch(M, C, x.device)),
               L.stream_ptr())
        ctx.save_for_backward(x, g, b, mean, invstd)
        ctx.act = int(act)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, g, b, mean, invstd = ctx.saved_tensors
        dy = dy.contiguous()
        C = x.shape[-1]
        M = x.numel() // C
        dx = torch.empty_like(x)
        dg = torch.empty(C, device=x.device)
        db = torch.empty(C, device=x.device)
        L.call("aql_bn_train_bwd", L.ptr(x), L.ptr(dy), L.ptr(g), L.ptr(b), L.ptr(mean), L.ptr(invstd), M, C, ctx.act,
               L.ptr(dx), L.ptr(dg), L.ptr(db), L.ptr(_bn_scratch(M, C, x.device)), L.stream_ptr())
        return dx, dg, db, None, None, None, None, None


class _BNResFn(torch.autograd.Function):
    """The last BatchNorm of an MBConv block with a skip connection, the stochastic-depth scale and the skip folded into its apply pass
    (torchvision MBConv.forward: ``result = stochastic_depth(block(input)); result += input``):  y = noise[b] * BN_train(x) + res.
    One launch set (statistics, finalize, apply) where BatchNorm + chan-scale + add ran; backward: d(res) = dy, and the BatchNorm
    backward scales dy per sample on the fly (aql_bn_train_fwd_res / aql_bn_train_bwd_rs, round 6)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, run_mean, run_var, eps, momentum, res, noise):
        x, res = x.contiguous(), res.contiguous()
        C = x.shape[-1]
        M = x.numel() // C
        rps = M // x.shape[0]
        noise = noise.to(x.device, torch.float32).contiguous()
        y = torch.empty_like(x)
        mean = torch.empty(C, device=x.device)
        invstd = torch.empty(C, device=x.device)
        g, b = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        L.call("aql_bn_train_fwd_res", L.ptr(x), L.ptr(g), L.ptr(b), M, C, float(eps), float(momentum), 0, L.ptr(res), L.ptr(noise), rps,
               L.ptr(y), L.ptr(mean), L.ptr(invstd), L.ptr(run_mean), L.ptr(run_var), L.ptr(_bn_scratch(M, C, x.device)), L.stream_ptr())
        ctx.save_for_backward(x, g, b, mean, invstd, noise)
        ctx.rps = rps
        return y

    @staticmethod
    def backward(ctx, dy):
        x, g, b, mean, invstd, noise = ctx.saved_tensors
        dy = dy.contiguous()
        C = x.shape[-1]
        M = x.numel() // C
        dx = torch.empty_like(x)
        dg = torch.empty(C, device=x.device)
        db = torch.empty(C, device=x.device)
        L.call("aql_bn_train_bwd_rs", L.ptr(x), L.ptr(dy), L.ptr(g), L.ptr(b), L.ptr(mean), L.ptr(invstd), M, C, 0, L.ptr(noise), ctx.rps,
               L.ptr(dx), L.ptr(dg), L.ptr(db), L.ptr(_bn_scratch(M, C, x.device)), L.stream_ptr())
        return dx, dg, db, None, None, None, None, dy, None


FUSE_BN_RES = True    # False = BatchNorm, chan-scale and residual add as three launch sets (module attribute: the parity test flips it)


def _bn_res(x, bn, res, noise):
    if bn.num_batches_tracked is not None:
        bn.num_batches_tracked += 1
    return _BNResFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps,
                          0.1 if bn.momentum is None else bn.momentum, res, noise)


def _bn_act(x, bn, act):
    if bn.num_batches_tracked is not None:
        bn.num_batches_tracked += 1
    return _BNActFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps,
                          0.1 if bn.momentum is None else bn.momentum, act)


class _DwConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, k, stride):
        x = x.contiguous()
        B, H, W, C = x.shape
        wp = w.detach().float().reshape(C, k * k).t().contiguous()       # [k*k][C]
        Ho = (H + 2 * (k // 2) - k) // stride + 1
        y = torch.empty(B, Ho, Ho, C, device=x.device)
        L.call("aql_dwconv_train", L.ptr(x), None, L.ptr(wp), B, H, W, C, k, stride, 0, L.ptr(y), L.stream_ptr())
        ctx.save_for_backward(x, wp)
        ctx.cfg = (k, stride, w.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wp = ctx.saved_tensors
        k, stride, wshape = ctx.cfg
        dy = dy.contiguous()
        B, H, W, C = x.shape
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            L.call("aql_dwconv_train", L.ptr(dy), None, L.ptr(wp), B, H, W, C, k, stride, 1, L.ptr(dx), L.stream_ptr())
        if ctx.needs_input_grad[1]:
            dwp = torch.empty(k * k, C, device=x.device)
            L.call("aql_dwconv_train", L.ptr(x), L.ptr(dy), None, B, H, W, C, k, stride, 2, L.ptr(dwp), L.stream_ptr())
            dw = dwp.t().reshape(wshape)
        return dx, dw, None, None


class _StemFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        B, H, W, _ = x.shape
        Co = w.shape[0]
        wp = w.detach().float().permute(2, 3, 1, 0).reshape(27, Co).contiguous()
        y = torch.empty(B, H // 2, W // 2, Co, device=x.device)
        L.call("aql_stem_train", L.ptr(x), None, L.ptr(wp), B, H, W, Co, 0, L.ptr(y), L.stream_ptr())
        ctx.save_for_backward(x, wp)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wp = ctx.saved_tensors
        dy = dy.contiguous()
        B, H, W, _ = x.shape
        Co = wp.shape[1]
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            L.call("aql_stem_train", L.ptr(dy), None, L.ptr(wp), B, H, W, Co, 1, L.ptr(dx), L.stream_ptr())
        if ctx.needs_input_grad[1]:
            dwp = torch.empty(27, Co, device=x.device)
            L.call("aql_stem_train", L.ptr(x), L.ptr(dy), None, B, H, W, Co, 2, L.ptr(dwp), L.stream_ptr())
            dw = dwp.view(3, 3, 3, Co).permute(3, 2, 0, 1).contiguous()
        return dx, dw


class _ChanScaleFn(torch.autograd.Function):
    """y[b,p,c] = x[b,p,c] * g[b,c]  (squeeze-excite gate, stochastic depth, dropout)."""

    @staticmethod
    def forward(ctx, x, g):
        x, g = x.contiguous(), g.contiguous()
        B, HW, C = x.shape
        y = torch.empty_like(x)
        L.call("aql_chan_scale", L.ptr(x), L.ptr(g), B, HW, C, L.ptr(y), L.stream_ptr())
        ctx.save_for_backward(x, g)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, g = ctx.saved_tensors
        dy = dy.contiguous()
        B, HW, C = x.shape
        dx = dg = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            L.call("aql_chan_scale", L.ptr(dy), L.ptr(g), B, HW, C, L.ptr(dx), L.stream_ptr())
        if ctx.needs_input_grad[1]:
            dg = torch.empty_like(g)
            L.call("aql_chan_reduce", L.ptr(dy), L.ptr(x), B, HW, C, 1.0, L.ptr(dg), L.stream_ptr())
        return dx, dg


class _ChanReduceFn(torch.autograd.Function):
    """global average pool [B,HW,C] -> [B,C]."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        B, HW, C = x.shape
        out = torch.empty(B, C, device=x.device)
        L.call("aql_chan_reduce", L.ptr(x), None, B, HW, C, 1.0 / HW, L.ptr(out), L.stream_ptr())
        ctx.shape = (B, HW, C)
        return out

    @staticmethod
    def backward(ctx, dy):
        B, HW, C = ctx.shape
        dx = torch.empty(B, HW, C, device=dy.device)
        L.call("aql_chan_bcast", L.ptr(dy.contiguous()), B, HW, C, 1.0 / HW, 0, L.ptr(dx), L.stream_ptr())
        return dx


class _ActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, kind):
        x = x.contiguous()
        y = torch.empty_like(x)
        L.call("aql_act_f32", L.ptr(x), None, kind, x.numel(), L.ptr(y), L.stream_ptr())
        ctx.save_for_backward(x)
        ctx.kind = kind
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dx = torch.empty_like(x)
        L.call("aql_act_f32", L.ptr(x), L.ptr(dy.contiguous()), ctx.kind, x.numel(), L.ptr(dx), L.stream_ptr())
        return dx, None


class _ResizeFn(torch.autograd.Function):
    """F.interpolate(x, (Ho,Wo), mode="bilinear") with NCHW in, NHWC out (models.py:92-94)."""

    @staticmethod
    def forward(ctx, x, Ho, Wo):
        B, C, H, W = x.shape
        y = torch.empty(B, Ho, Wo, C, device=x.device)
        L.call("aql_resize_bilinear_nhwc", L.ptr(x), B, C, H, W, Ho, Wo, L.ptr(y), L.stream_ptr())
        ctx.cfg = (B, C, H, W, Ho, Wo)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, C, H, W, Ho, Wo = ctx.cfg
        dx = torch.empty(B, C, H, W, device=dy.device)
        L.call("aql_resize_bilinear_nhwc_bwd", L.ptr(dy.contiguous()), B, C, H, W, Ho, Wo, L.ptr(dx), L.stream_ptr())
        return dx, None, None


class _BceFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target):
        z, t = logits.float().contiguous(), target.float().contiguous()
        loss = torch.empty((), device=z.device)
        dz = torch.empty_like(z)
        L.call("aql_bce_logits", L.ptr(z), L.ptr(t), z.numel(), L.ptr(loss), L.ptr(dz), L.stream_ptr())
        ctx.save_for_backward(dz)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dz,) = ctx.saved_tensors
        return dz * g, None


def bce_with_logits(logits, target):
    """F.binary_cross_entropy_with_logits(logits, target) (mean reduction), latent_wm_pretrain.py:196."""
    if not logits.is_cuda:
        raise L.AqlError("bce_with_logits: the HIP path needs GPU tensors; there is no CPU fallback")
    return _BceFn.apply(logits, target)
