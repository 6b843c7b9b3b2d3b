"""SecretDecoder (reference utils/models.py:84-96; duplicate evaluation/utils_eval.py:142-154): torchvision
``efficientnet_b1`` with ``classifier[1] = Linear(1280, 2*bits)``; ``forward(x)`` resizes to 512x512 (bilinear) and
returns logits ``[B, bits, 2]``; callers take ``argmax(-1)`` as the message bits.

torchvision is not installed on the target image, so the module tree below restates torchvision 0.15's public
EfficientNet-B1 layout (stem 3->32 s2; MBConv stages (t,k,s,c,n) = (1,3,1,16,2) (6,3,2,24,3) (6,5,2,40,3) (6,3,2,80,4)
(6,5,1,112,4) (6,5,2,192,5) (6,3,1,320,2); SE squeeze = block-input/4; head 320->1280; BN eps 1e-5) with the same
state-dict key names as ``msgdecoder.pt`` (``model.features.*``, ``model.classifier.1.*``).  PARITY UNPINNED against
torchvision itself: HIP is checked against oracle/decoder_oracle.py, a CPU restatement of the same
public architecture.

Inference (eval mode: BN running stats, no dropout / stochastic depth) runs in the fp32 HIP kernels of
csrc/aql_decoder.hip.  Training mode (stage 1 / rob-finetune) is not built yet and raises.
"""
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
        if mode:
            raise NotImplementedError("SecretDecoder training (BN batch statistics, backward) is not built yet; "
                                      "inference only (call .eval())")
        return super().train(False)

    # ---------------------------------------------------------------------------------------- forward
    @torch.no_grad()
    def forward(self, x):
        if not x.is_cuda:
            raise L.AqlError("SecretDecoder: the HIP path needs GPU tensors; there is no CPU fallback")
        P = self._packed or self._pack()
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
            pool = torch.empty(B, cexp, device=dev)
            L.call("aql_avgpool_nhwc", L.ptr(dw), B, Hc * Wc, cexp, L.ptr(pool), st)
            gate = torch.empty(B, cexp, device=dev)
            w1, b1, w2, b2 = d["se"]
            L.call("aql_se_gate", L.ptr(pool), L.ptr(w1), L.ptr(b1), L.ptr(w2), L.ptr(b2), B, cexp, w1.shape[0],
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
