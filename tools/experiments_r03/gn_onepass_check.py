"""GPU test of the shelved one-pass GroupNorm (tools/experiments_r03/gn_onepass.cuh); passed on MI355X before the kernel was shelved."""
import torch


def test_groupnorm_onepass_vs_torch_and_two_pass_forms():
    """aql_groupnorm_silu_fwd/_bwd_onepass (rows in registers, in-kernel barrier per sample) on the U-Net's GroupNorm shapes (twin
    batch 8 forward, batch 4 backward, ragged / non-square maps, concat widths) against fp32 torch and the two-pass kernels; two
    launches give identical bits (fixed summation order); the barrier counters are left zero and the error flag stays down."""
    import torch.nn.functional as F
    from aqualora_amd import _lib as L
    torch.manual_seed(3)
    served = 0
    for B, C, H, W, silu in ((8, 320, 64, 64, 1), (4, 320, 64, 64, 1), (8, 640, 32, 32, 1), (4, 960, 64, 64, 0), (8, 1280, 16, 16, 1),
                             (4, 2560, 8, 8, 1), (4, 1920, 16, 16, 1), (3, 320, 72, 88, 1), (2, 640, 9, 7, 0), (8, 1280, 8, 8, 1)):
        x = (torch.randn(B, H, W, C, device="cuda") * 2 + 0.5).to(torch.bfloat16)
        dy = torch.randn(B, H, W, C, device="cuda").to(torch.bfloat16)
        dres = torch.randn(B, H, W, C, device="cuda").to(torch.bfloat16)
        ga = (torch.randn(C, device="cuda") * 0.5 + 1).to(torch.bfloat16)
        be = (torch.randn(C, device="cuda") * 0.1).to(torch.bfloat16)
        sync = torch.zeros(16 * 256 * 64 + 16 * 2 + 1, dtype=torch.int32, device="cuda")
        scr = torch.zeros(1 << 18, device="cuda")
        outs = []
        for rep in range(2):
            y = torch.full_like(x, float("nan"))
            stats = torch.full((B, 32, 2), float("nan"), device="cuda")
            rc = L.call_raw("aql_groupnorm_silu_fwd_onepass", L.ptr(x), B, H * W, C, L.ptr(ga), L.ptr(be), 1e-5, silu, L.ptr(y),
                            L.ptr(stats), L.ptr(sync), L.stream_ptr())
            if rc == 100:
                break
            assert rc == 0
            dx = torch.full_like(x, float("nan"))
            rc = L.call_raw("aql_groupnorm_silu_bwd_onepass", L.ptr(x), L.ptr(dy), B, H * W, C, L.ptr(ga), L.ptr(be), silu,
                            L.ptr(stats), L.ptr(dres), L.ptr(dx), L.ptr(sync), L.stream_ptr())
            assert rc in (0, 100)
            outs.append((y, stats, dx if rc == 0 else None))
        if not outs:
            continue
        served += 1
        assert int(sync[16 * 256 * 64:].abs().max()) == 0, (B, C, H, W)     # counters reset, error flag down
        (y, stats, dx), (y2, stats2, dx2) = outs
        assert torch.equal(y, y2) and torch.equal(stats, stats2) and (dx is None or torch.equal(dx, dx2)), (B, C, H, W)
        xr = x.float().permute(0, 3, 1, 2).requires_grad_(True)
        yr = F.group_norm(xr, 32, ga.float(), be.float(), 1e-5)
        yr = F.silu(yr) if silu else yr
        err = float((y.float().permute(0, 3, 1, 2) - yr).abs().max() / yr.abs().max())
        assert err < 1.5e-2, (B, C, H, W, err)
        y0, st0 = torch.empty_like(x), torch.empty(B, 32, 2, device="cuda")
        L.call("aql_groupnorm_silu_fwd", L.ptr(x), B, H * W, C, L.ptr(ga), L.ptr(be), 1e-5, silu, L.ptr(y0), L.ptr(st0), L.ptr(scr),
               L.stream_ptr())
        assert float((stats - st0).abs().max() / st0.abs().max()) < 1e-5, (B, C, H, W)
        assert float((y.float() - y0.float()).abs().max() / y0.float().abs().max()) < 8e-3, (B, C, H, W)   # a bf16 ulp at boundaries
        if dx is not None:
            yr.backward(dy.float().permute(0, 3, 1, 2))
            dxr = xr.grad + dres.float().permute(0, 3, 1, 2)
            err = float((dx.float().permute(0, 3, 1, 2) - dxr).abs().max() / dxr.abs().max())
            assert err < 2e-2, (B, C, H, W, err)
    assert served >= 8


