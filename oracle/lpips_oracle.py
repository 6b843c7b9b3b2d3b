"""TEST INFRASTRUCTURE (CPU oracle, never imported by the product path): plain-PyTorch fp32 restatement of
``lpips.LPIPS(net='vgg')`` as called at train/latent_wm_pretrain.py:111,182.  The lpips package (0.1.4) is third-party and
not on disk, so this follows its published algorithm -- ScalingLayer, torchvision VGG16 feature slices up to relu1_2 /
relu2_2 / relu3_3 / relu4_3 / relu5_3, normalize_tensor (eps 1e-10), squared difference, 1x1 linear heads, spatial mean, sum
over the five taps -- PARITY UNPINNED (no reference vectors exist for it)."""
import torch
import torch.nn.functional as F

SLICES = (((0, 3, 64), (2, 64, 64)), ((5, 64, 128), (7, 128, 128)), ((10, 128, 256), (12, 256, 256), (14, 256, 256)),
          ((17, 256, 512), (19, 512, 512), (21, 512, 512)), ((24, 512, 512), (26, 512, 512), (28, 512, 512)))


def lpips_vgg(sd, in0, in1, bf16=False):
    """-> [B,1,1,1].  ``bf16=True`` mirrors the HIP path's rounding points (bf16 weights and activations, fp32 accumulate)."""
    r = (lambda t: t.to(torch.bfloat16).float()) if bf16 else (lambda t: t)
    shift = torch.tensor([-.030, -.088, -.188]).view(1, 3, 1, 1)
    scale = torch.tensor([.458, .448, .450]).view(1, 3, 1, 1)

    def feats(x):
        h = r((x - shift) / scale)
        out = []
        for s, convs in enumerate(SLICES, start=1):
            if s > 1:
                h = F.max_pool2d(h, 2, 2)
            for idx, _, _ in convs:
                h = r(F.relu(r(F.conv2d(h, r(sd[f"net.slice{s}.{idx}.weight"]), r(sd[f"net.slice{s}.{idx}.bias"]), padding=1))))
            out.append(h)
        return out

    f0, f1 = feats(in0.float()), feats(in1.float())
    total = 0
    for i, (a, b) in enumerate(zip(f0, f1)):
        na = a / (a.pow(2).sum(1, keepdim=True).sqrt() + 1e-10)
        nb = b / (b.pow(2).sum(1, keepdim=True).sqrt() + 1e-10)
        d = F.conv2d((na - nb) ** 2, sd[f"lin{i}.model.1.weight"].float())
        total = total + d.mean(dim=(2, 3), keepdim=True)
    return total
