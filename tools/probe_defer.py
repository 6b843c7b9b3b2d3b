"""GPU probe (round 6): a split-K 3x3 convolution whose finalize launch is left to the GroupNorm behind it.
  (1) kernel level: aql_conv3x3_fwd_defer + aql_groupnorm_silu_fwd_slabs  ==  aql_conv3x3_fwd + aql_groupnorm_silu_fwd   (conv output,
      normalised map, statistics: BIT for bit), with bias / per-sample row bias / residual epilogues, on the 8x8 .. 32x32 maps of the U-Net
      at sampling (2), training (4 / 8) batch sizes; backward: aql_conv3x3_bwd_data_defer + aql_groupnorm_silu_bwd_slabs == the two launches;
  (2) aql_splitk_finalize after a deferred launch == the undeferred convolution;
  (3) module level: ResnetBlock2D forward + backward with ops.DEFER_FINALIZE on == off (outputs and input gradients bit for bit), and a
      deferred output read by another op first is finished by that op (ops._req).
Prints PASS / FAIL per case and ALL PASS."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aqualora_amd import _lib as L, ops, synth   # noqa: E402

dev = torch.device("cuda", 0)
CL = torch.channels_last
ok_all = True
n_fused = 0


def say(ok, msg):
    global ok_all
    ok_all &= bool(ok)
    print(("PASS " if ok else "FAIL ") + msg)


def rnd(name, shape, std=1.0):
    return synth.normal(name, shape, std, 77, dev)


def conv_case(B, H, C_in, C_out, rowbias, residual, silu=1):
    global n_fused
    tag = f"B={B} {H}x{H} {C_in}->{C_out} rowbias={int(rowbias)} residual={int(residual)}"
    x = rnd(tag + "x", (B, H, H, C_in)).to(torch.bfloat16)
    wk = rnd(tag + "w", (C_out, 9 * C_in), (9 * C_in) ** -0.5).to(torch.bfloat16)
    bias = rnd(tag + "b", (C_out,), 0.1).to(torch.bfloat16)
    rb = rnd(tag + "rb", (B, C_out), 0.5).to(torch.bfloat16) if rowbias else None
    res = rnd(tag + "res", (B, H, H, C_out)).to(torch.bfloat16) if residual else None
    gamma = (1 + rnd(tag + "g", (C_out,), 0.1)).to(torch.bfloat16)
    beta = rnd(tag + "be", (C_out,), 0.1).to(torch.bfloat16)
    ws = torch.empty(64 << 20, dtype=torch.float32, device=dev)
    scr = torch.empty(1 << 20, dtype=torch.float32, device=dev)
    st = L.stream_ptr()

    def ref():
        y = torch.full((B, H, H, C_out), float("nan"), dtype=torch.bfloat16, device=dev)
        n = torch.empty_like(y)
        stats = torch.empty(B, 32, 2, dtype=torch.float32, device=dev)
        L.call("aql_conv3x3_fwd", L.ptr(x), B, H, H, C_in, L.ptr(wk), L.ptr(bias), C_out, 1, 0, L.ptr(rb), 0 if rb is None else C_out,
               L.ptr(res), L.ptr(y), L.ptr(ws), ws.numel() * 4, st)
        L.call("aql_groupnorm_silu_fwd", L.ptr(y), B, H * H, C_out, L.ptr(gamma), L.ptr(beta), 1e-5, silu, L.ptr(n), L.ptr(stats),
               L.ptr(scr), st)
        return y, n, stats

    y0, n0, s0 = ref()
    y = torch.full((B, H, H, C_out), float("nan"), dtype=torch.bfloat16, device=dev)
    n = torch.empty_like(y)
    stats = torch.empty(B, 32, 2, dtype=torch.float32, device=dev)
    ns = ctypes.c_int(0)
    L.call("aql_conv3x3_fwd_defer", L.ptr(x), B, H, H, C_in, L.ptr(wk), L.ptr(bias), C_out, 1, 0, L.ptr(rb), 0 if rb is None else C_out,
           L.ptr(res), L.ptr(y), L.ptr(ws), ws.numel() * 4, ctypes.byref(ns), st)
    how = f"splits {ns.value}"
    if ns.value > 1:
        # (2) the finalize on its own, into a second buffer
        y2 = torch.empty_like(y)
        L.call("aql_splitk_finalize", L.ptr(ws), ns.value, B * H * H, C_out, L.ptr(bias), L.ptr(rb), C_out, H * H, L.ptr(res), C_out,
               L.ptr(y2), C_out, st)
        say(torch.equal(y2.view(torch.int16), y0.view(torch.int16)), f"aql_splitk_finalize after a deferred launch  {tag}  {how}")
        rc = L.call_raw("aql_groupnorm_silu_fwd_slabs", L.ptr(ws), ns.value, L.ptr(bias), L.ptr(rb), C_out, L.ptr(res), L.ptr(y), B, H * H,
                        C_out, L.ptr(gamma), L.ptr(beta), 1e-5, silu, L.ptr(n), L.ptr(stats), st)
        if rc == 100:
            say(True, f"forward {tag} {how}: the one-launch GroupNorm does not take this map (caller finalizes)")
            return
        L.check(rc, "aql_groupnorm_silu_fwd_slabs")
        n_fused += 1
    else:
        L.call("aql_groupnorm_silu_fwd", L.ptr(y), B, H * H, C_out, L.ptr(gamma), L.ptr(beta), 1e-5, silu, L.ptr(n), L.ptr(stats),
               L.ptr(scr), st)
    torch.cuda.synchronize()
    e = (torch.equal(y.view(torch.int16), y0.view(torch.int16)), torch.equal(n.view(torch.int16), n0.view(torch.int16)), torch.equal(stats, s0))
    say(all(e), f"forward  {tag}  {how}: conv output / normalised map / statistics bit-equal {e}")


def bwd_case(B, H, C_in, C_out, dres, silu=1):
    """conv: C_in -> C_out; backward-data produces d(conv input) [B,H,H,C_in], consumed by the GroupNorm backward over C_in channels"""
    global n_fused
    tag = f"B={B} {H}x{H} {C_in}<-{C_out} dres={int(dres)}"
    dy = rnd(tag + "dy", (B, H, H, C_out)).to(torch.bfloat16)
    wt = rnd(tag + "wt", (C_in, 9 * C_out), (9 * C_out) ** -0.5).to(torch.bfloat16)
    xg = rnd(tag + "xg", (B, H, H, C_in)).to(torch.bfloat16)        # the GroupNorm's saved input
    gamma = (1 + rnd(tag + "g", (C_in,), 0.1)).to(torch.bfloat16)
    beta = rnd(tag + "be", (C_in,), 0.1).to(torch.bfloat16)
    dr = rnd(tag + "dr", (B, H, H, C_in)).to(torch.bfloat16) if dres else None
    ws = torch.empty(64 << 20, dtype=torch.float32, device=dev)
    scr = torch.empty(1 << 20, dtype=torch.float32, device=dev)
    st = L.stream_ptr()
    stats = torch.empty(B, 32, 2, dtype=torch.float32, device=dev)
    tmp = torch.empty_like(xg)
    L.call("aql_groupnorm_silu_fwd", L.ptr(xg), B, H * H, C_in, L.ptr(gamma), L.ptr(beta), 1e-5, silu, L.ptr(tmp), L.ptr(stats), L.ptr(scr), st)
    dn0 = torch.empty_like(xg)
    dx0 = torch.empty_like(xg)
    L.call("aql_conv3x3_bwd_data", L.ptr(dy), B, H, H, C_in, L.ptr(wt), C_out, 1, L.ptr(dn0), L.ptr(ws), ws.numel() * 4, st)
    L.call("aql_groupnorm_silu_bwd", L.ptr(xg), L.ptr(dn0), B, H * H, C_in, L.ptr(gamma), L.ptr(beta), silu, L.ptr(stats), L.ptr(dr),
           L.ptr(dx0), L.ptr(scr), st)
    dn = torch.full_like(xg, float("nan"))
    dx = torch.empty_like(xg)
    ns = ctypes.c_int(0)
    L.call("aql_conv3x3_bwd_data_defer", L.ptr(dy), B, H, H, C_in, L.ptr(wt), C_out, 1, L.ptr(dn), L.ptr(ws), ws.numel() * 4,
           ctypes.byref(ns), st)
    how = f"splits {ns.value}"
    if ns.value > 1:
        rc = L.call_raw("aql_groupnorm_silu_bwd_slabs", L.ptr(xg), L.ptr(ws), ns.value, B, H * H, C_in, L.ptr(gamma), L.ptr(beta), silu,
                        L.ptr(stats), L.ptr(dr), L.ptr(dx), st)
        if rc == 100:
            say(True, f"backward {tag} {how}: the one-launch GroupNorm backward does not take this map (caller finalizes)")
            return
        L.check(rc, "aql_groupnorm_silu_bwd_slabs")
        n_fused += 1
    else:
        L.call("aql_groupnorm_silu_bwd", L.ptr(xg), L.ptr(dn), B, H * H, C_in, L.ptr(gamma), L.ptr(beta), silu, L.ptr(stats), L.ptr(dr),
               L.ptr(dx), L.ptr(scr), st)
    torch.cuda.synchronize()
    say(torch.equal(dx.view(torch.int16), dx0.view(torch.int16)), f"backward {tag}  {how}: GroupNorm input gradient bit-equal")


def block_case(B, H, C_in, C_out, twin):
    from aqualora_amd.unet import ResnetBlock2D
    tag = f"ResnetBlock2D B={B} {H}x{H} {C_in}->{C_out} twin={int(twin)}"
    torch.manual_seed(5)
    blk = ResnetBlock2D(C_in, C_out, 1280, 1e-5, device=dev, dtype=torch.bfloat16)
    with torch.no_grad():
        for n_, p_ in blk.named_parameters():
            p_.copy_(rnd(tag + n_, tuple(p_.shape), 0.05 if p_.dim() > 1 else 0.3).to(p_.dtype))
            if n_.endswith("norm1.weight") or n_.endswith("norm2.weight"):
                p_.add_(1.0)
    x0 = rnd(tag + "x", (2 * B if twin else B, C_in, H, H)).to(torch.bfloat16).contiguous(memory_format=CL)
    temb = rnd(tag + "t", (2 * B if twin else B, 1280)).to(torch.bfloat16)
    gy = rnd(tag + "gy", (B, C_out, H, H)).to(torch.bfloat16).contiguous(memory_format=CL)
    outs = {}
    for mode in (False, True):
        ops.DEFER_FINALIZE = mode
        if twin:
            ops.dual_begin()
            ops.DUAL.register(x0)
            x = x0[B:].detach().requires_grad_(True)
        else:
            x = x0.detach().clone(memory_format=torch.preserve_format).requires_grad_(True)
        try:
            y = blk(x, temb, None)
        finally:
            if twin:
                ops.dual_end()
        y.backward(gy)
        torch.cuda.synchronize()
        outs[mode] = (y.detach().clone(), x.grad.detach().clone())
        assert not ops._PENDING, "a deferred finalize was left unconsumed"
    ops.DEFER_FINALIZE = True
    e = (torch.equal(outs[False][0].view(torch.int16), outs[True][0].view(torch.int16)),
         torch.equal(outs[False][1].view(torch.int16), outs[True][1].view(torch.int16)))
    say(all(e), f"{tag}: output / input gradient bit-equal with the finalize launches held back {e}")


def reader_case():
    """a deferred convolution output read by an op other than the GroupNorm: ops._req finishes it first"""
    from aqualora_amd.unet import ResnetBlock2D
    from aqualora_amd.lora import _packed_conv3
    B, H, C = 2, 16, 1280
    blk = ResnetBlock2D(C, C, 1280, 1e-5, device=dev, dtype=torch.bfloat16)
    with torch.no_grad():
        blk.conv1.weight.copy_(rnd("rd.w", tuple(blk.conv1.weight.shape), 0.01).to(torch.bfloat16))
    x = rnd("rd.x", (B, C, H, H)).to(torch.bfloat16).contiguous(memory_format=CL)
    with torch.no_grad():
        ops.DEFER_FINALIZE = False
        y0 = ops.conv3x3(x, _packed_conv3(blk.conv1), False, None, None, gn_next=True)
        ops.DEFER_FINALIZE = True
        y1 = ops.conv3x3(x, _packed_conv3(blk.conv1), False, None, None, gn_next=True)
        pending = bool(ops._PENDING)
        # a second deferred convolution asks for the slab buffer: the first one is finished before its slabs are overwritten
        y2 = ops.conv3x3(x, _packed_conv3(blk.conv1), False, None, None, gn_next=True)
        ops.flush_all_pending()
    torch.cuda.synchronize()
    say(pending and torch.equal(y1.view(torch.int16), y0.view(torch.int16)) and torch.equal(y2.view(torch.int16), y0.view(torch.int16)),
        f"a deferred output (pending={pending}) is finished by the next user of the slab buffer / flush_all_pending")


if __name__ == "__main__":
    torch.cuda.set_device(0)
    for B in (2, 8):
        for (H, ci, co) in ((8, 1280, 1280), (16, 1280, 1280), (16, 640, 1280), (32, 640, 640), (32, 320, 640), (8, 2560, 1280)):
            conv_case(B, H, ci, co, rowbias=True, residual=False)
    conv_case(2, 64, 320, 320, True, False)
    conv_case(2, 16, 1280, 1280, False, True)
    conv_case(4, 8, 1280, 1280, True, True, silu=0)
    conv_case(1, 16, 1280, 1280, True, False)
    for B in (2, 4, 8):
        for (H, ci, co) in ((8, 1280, 1280), (16, 1280, 1280), (16, 640, 1280), (32, 640, 640), (32, 1920, 640), (8, 2560, 1280)):
            bwd_case(B, H, ci, co, dres=(H == 16))
    block_case(4, 16, 1280, 1280, twin=True)
    block_case(4, 8, 2560, 1280, twin=True)
    block_case(4, 32, 640, 640, twin=False)
    block_case(2, 16, 640, 1280, twin=False)
    reader_case()
    print(f"launches whose finalize ran inside the GroupNorm: {n_fused}")
    print("ALL PASS" if ok_all and n_fused >= 20 else "SOME FAILED")
