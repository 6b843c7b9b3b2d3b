"""ORACLE -- test infrastructure only.  CPU restatement of SecretDecoder.forward (reference utils/models.py:91-96):
bilinear resize to 512x512, torchvision EfficientNet-B1 in eval mode, Linear(1280, 2*bits), view(-1, bits, 2).

PARITY UNPINNED: torchvision==0.15.2 (requirements.txt:30) is not on disk in this container and the reference holds
no test vectors for the decoder, so this restates torchvision's PUBLIC architecture (Conv2dNormActivation = conv +
BatchNorm2d(eps 1e-5) + SiLU; MBConv = [expand 1x1] -> depthwise kxk -> SqueezeExcitation(avgpool, fc1, SiLU, fc2,
sigmoid, scale) -> project 1x1 [+ residual]; stochastic depth / dropout are identities in eval) over a state dict
with torchvision's key names.  It pins the HIP kernels to plain PyTorch ops, not to torchvision's implementation.
"""
import torch
import torch.nn.functional as F

B1_STAGES = [(1, 3, 1, 32, 16, 2), (6, 3, 2, 16, 24, 3), (6, 5, 2, 24, 40, 3), (6, 3, 2, 40, 80, 4),
             (6, 5, 1, 80, 112, 4), (6, 5, 2, 112, 192, 5), (6, 3, 1, 192, 320, 2)]


def _cna(sd, p, x, stride, k, groups=1, act=True):
    x = F.conv2d(x, sd[p + ".0.weight"], None, stride, (k - 1) // 2, 1, groups)
    x = F.batch_norm(x, sd[p + ".1.running_mean"], sd[p + ".1.running_var"], sd[p + ".1.weight"], sd[p + ".1.bias"],
                     False, 0.0, 1e-5)
    return F.silu(x) if act else x


def secret_decoder(sd, x, bits):
    sd = {k[len("model."):] if k.startswith("model.") else k: v.float() for k, v in sd.items()}
    x = F.interpolate(x.float(), size=(512, 512), mode="bilinear")
    x = _cna(sd, "features.0", x, 2, 3)
    for si, (t, k, s, cin, cout, n) in enumerate(B1_STAGES, start=1):
        for i in range(n):
            p = f"features.{si}.{i}.block"
            stride = s if i == 0 else 1
            c_in = cin if i == 0 else cout
            inp = x
            j = 0
            if t != 1:
                x = _cna(sd, f"{p}.0", x, 1, 1)
                j = 1
            x = _cna(sd, f"{p}.{j}", x, stride, k, groups=x.shape[1])
            g = x.mean(dim=(2, 3), keepdim=True)
            g = F.silu(F.conv2d(g, sd[f"{p}.{j + 1}.fc1.weight"], sd[f"{p}.{j + 1}.fc1.bias"]))
            g = torch.sigmoid(F.conv2d(g, sd[f"{p}.{j + 1}.fc2.weight"], sd[f"{p}.{j + 1}.fc2.bias"]))
            x = x * g
            x = _cna(sd, f"{p}.{j + 2}", x, 1, 1, act=False)
            if stride == 1 and c_in == cout:
                x = x + inp
    x = _cna(sd, "features.8", x, 1, 1)
    x = x.mean(dim=(2, 3))
    x = F.linear(x, sd["classifier.1.weight"], sd["classifier.1.bias"])
    return x.view(-1, bits, 2)


def _cna_train(sd, bufs, p, x, stride, k, groups=1, act=True):
    """Conv2dNormActivation in train() mode: batch statistics, running stats updated in ``bufs`` (momentum 0.1)."""
    x = F.conv2d(x, sd[p + ".0.weight"], None, stride, (k - 1) // 2, 1, groups)
    x = F.batch_norm(x, bufs[p + ".1.running_mean"], bufs[p + ".1.running_var"], sd[p + ".1.weight"], sd[p + ".1.bias"],
                     True, 0.1, 1e-5)
    return F.silu(x) if act else x


def secret_decoder_train(sd, x, bits, sd_noise, drop_mask):
    """train()-mode forward (latent_wm_pretrain.py:160 ``sec_decoder.train()``): BatchNorm batch statistics, torchvision
    StochasticDepth("row") with the per-sample factors ``sd_noise[block]`` ([B], already / survival), Dropout(0.2) with
    the given mask ([B,1280], already / 0.8).  ``sd`` holds (possibly requires_grad) parameters keyed like msgdecoder.pt
    without the ``model.`` prefix; running statistics are read from and updated in ``sd`` as well (plain tensors)."""
    bufs = sd
    x = F.interpolate(x.float(), size=(512, 512), mode="bilinear")
    x = _cna_train(sd, bufs, "features.0", x, 2, 3)
    bi = 0
    for si, (t, k, s, cin, cout, n) in enumerate(B1_STAGES, start=1):
        for i in range(n):
            p = f"features.{si}.{i}.block"
            stride = s if i == 0 else 1
            c_in = cin if i == 0 else cout
            inp = x
            j = 0
            if t != 1:
                x = _cna_train(sd, bufs, f"{p}.0", x, 1, 1)
                j = 1
            x = _cna_train(sd, bufs, f"{p}.{j}", x, stride, k, groups=x.shape[1])
            g = x.mean(dim=(2, 3), keepdim=True)
            g = F.silu(F.conv2d(g, sd[f"{p}.{j + 1}.fc1.weight"], sd[f"{p}.{j + 1}.fc1.bias"]))
            g = torch.sigmoid(F.conv2d(g, sd[f"{p}.{j + 1}.fc2.weight"], sd[f"{p}.{j + 1}.fc2.bias"]))
            x = x * g
            x = _cna_train(sd, bufs, f"{p}.{j + 2}", x, 1, 1, act=False)
            if stride == 1 and c_in == cout:
                x = x * sd_noise[bi].view(-1, 1, 1, 1) + inp
            bi += 1
    x = _cna_train(sd, bufs, "features.8", x, 1, 1)
    x = x.mean(dim=(2, 3)) * drop_mask
    x = F.linear(x, sd["classifier.1.weight"], sd["classifier.1.bias"])
    return x.view(-1, bits, 2)
