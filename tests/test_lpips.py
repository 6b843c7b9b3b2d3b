"""LPIPS(VGG16): CPU -- the key inventory and the oracle's basic properties; GPU -- HIP forward and image gradient against
the oracle (oracle/lpips_oracle.py, parity UNPINNED: the lpips package is absent from the image)."""
import pytest
import torch

from tests.common import T


def test_lpips_key_inventory_and_oracle_properties():
    from aqualora_amd.lpips import lpips_keys, synthetic_state_dict
    from oracle.lpips_oracle import lpips_vgg
    keys = lpips_keys()
    assert len(keys) == 13 * 2 + 5
    assert keys["net.slice1.0.weight"] == (64, 3, 3, 3) and keys["net.slice5.28.weight"] == (512, 512, 3, 3)
    assert keys["lin0.model.1.weight"] == (1, 64, 1, 1) and keys["lin4.model.1.weight"] == (1, 512, 1, 1)
    assert sum(torch.Size(s).numel() for s in keys.values()) == 14714688 + 64 + 128 + 256 + 512 + 512   # VGG16 convs + heads
    sd = synthetic_state_dict()
    a = T("lp.a", (2, 3, 32, 64), 0.4).clamp(-1, 1)
    b = (a + T("lp.d", (2, 3, 32, 64), 0.1)).clamp(-1, 1)
    with torch.no_grad():
        d_ab, d_aa, d_ba = lpips_vgg(sd, a, b), lpips_vgg(sd, a, a), lpips_vgg(sd, b, a)
    assert d_ab.shape == (2, 1, 1, 1) and float(d_aa.abs().max()) == 0.0
    assert torch.allclose(d_ab, d_ba, rtol=1e-5) and float(d_ab.min()) > 0        # symmetric, positive


@pytest.mark.gpu
def test_lpips_hip_vs_oracle():
    from aqualora_amd.lpips import LPIPS, synthetic_state_dict
    from oracle.lpips_oracle import lpips_vgg
    sd = synthetic_state_dict()
    net = LPIPS(sd, "cuda")
    for shape in ((2, 3, 64, 96), (1, 3, 128, 128)):
        a = T("lp.a" + str(shape), shape, 0.4).clamp(-1, 1)
        b = (a + T("lp.d" + str(shape), shape, 0.15)).clamp(-1, 1)
        bg = b.clone().cuda().requires_grad_(True)
        got = net(a.cuda(), bg)
        w = T("lp.w" + str(shape), (shape[0], 1, 1, 1)).abs() + 0.5
        (got * w.cuda()).sum().backward()
        br = b.clone().requires_grad_(True)
        want = lpips_vgg(sd, a, br, bf16=True)
        (want * w).sum().backward()
        want32 = lpips_vgg(sd, a, b)
        rel = ((got.cpu() - want).abs() / want.abs()).max().item()
        rel32 = ((got.cpu() - want32).abs() / want32.abs()).max().item()
        gl2 = ((bg.grad.cpu() - br.grad).norm() / br.grad.norm()).item()
        print(f"lpips {shape}: value vs bf16-mirroring oracle {rel:.2e}, vs fp32 {rel32:.2e}; image-gradient l2rel {gl2:.3f}")
        # measured on MI355X: 7e-5 / 3e-4 / 0.02
        assert rel < 1e-3 and rel32 < 5e-3, (rel, rel32)
        assert gl2 < 0.05, gl2
    with torch.no_grad():
        same = net(a.cuda(), a.cuda())
    assert float(same.abs().max()) < 1e-6
