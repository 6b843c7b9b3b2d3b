"""Host logic of the row-resident chains (aqualora_amd.ops.chain_ok / span_chunks): which launch sequences of the 320-channel level
are handed to aql_lora_chain_fwd / _r320 and which stay per-launch.  CPU only: no library call is made."""
import types

import torch

from aqualora_amd import ops


def _stage(N=320, K=320, rank=32):
    packed = types.SimpleNamespace(N=N, K=K, w=None, bias=None)
    site = None if rank is None else types.SimpleNamespace(rank=rank)
    return ops.ChainStage(packed, site, True)


def _x(M, C=320):
    return torch.empty(M, C, dtype=torch.bfloat16)


def test_chain_gate_takes_the_config_2_and_config_3_shapes():
    S32, S320 = torch.empty(8, 32, dtype=torch.bfloat16), torch.empty(16, 320, dtype=torch.bfloat16)
    # config 2: twin batch of 8 x 4096 tokens, rank 32 -- and its backward half
    assert ops.chain_ok(_x(32768), [_stage(), _stage()], S32, 4096)
    assert ops.chain_ok(_x(16384), [_stage()], S32, 4096)
    # config 3: twin batch of 16 x 4096 tokens, rank 320 (64-row tiles, one workgroup per CU: at least 256 tiles)
    assert ops.chain_ok(_x(65536), [_stage(rank=320)] * 4, S320, 4096)
    assert not ops.chain_ok(_x(8192), [_stage(rank=320)], S320[:2], 4096)
    # a rank without a chain kernel, mixed ranks, a scale whose width disagrees with the sites
    assert not ops.chain_ok(_x(32768), [_stage(rank=64)], torch.empty(8, 64, dtype=torch.bfloat16), 4096)
    assert not ops.chain_ok(_x(32768), [_stage(rank=32), _stage(rank=320)], S32, 4096)
    assert not ops.chain_ok(_x(65536), [_stage(rank=32)], S320, 4096)


def test_chain_gate_refuses_what_the_kernel_cannot_take():
    S32 = torch.empty(8, 32, dtype=torch.bfloat16)
    ok = [_stage()]
    assert not ops.chain_ok(_x(32768, 640), [_stage(640, 640)], S32, 1024)            # other widths: the tile does not fit the LDS
    assert not ops.chain_ok(_x(32768), [_stage(N=1280)], S32, 4096)                    # ff.net.0 / ff.net.2 are not 320 -> 320
    assert not ops.chain_ok(_x(32768 + 32), ok, S32, 4096)                             # whole 64-row tiles only
    assert not ops.chain_ok(_x(32768), ok, S32, 4096 + 32)                             # a tile may not straddle two samples
    assert not ops.chain_ok(_x(4096), ok, S32[:1], 4096)                               # too few tiles to fill half the chip
    assert not ops.chain_ok(_x(32768).float(), ok, S32, 4096)                          # bf16 only
    assert not ops.chain_ok(_x(32768), [_stage(rank=None)], S32, 4096)                 # a LoRA-free linear inside a LoRA chain
    # LoRA-free chains (fused weights / clean pass): only from 16384 rows on (measured neutral below: DESIGN section 4)
    assert ops.chain_ok(_x(16384), [_stage(rank=None)], None, 4096)
    assert not ops.chain_ok(_x(8192), [_stage(rank=None)], None, 4096)


def test_span_chunks_keeps_every_piece_under_the_descriptor_span():
    # (the kernels read operands through 1 GiB buffer descriptors: ops.span_chunks cuts rows / samples so that no piece reaches it)
    for n, per, align in ((16, 512 * 512 * 256 * 2, 1), (65536, 320 * 2, 64), (3, 1 << 29, 1), (1, 1 << 31, 1)):
        pieces = ops.span_chunks(n, per, align)
        assert sum(c for _, c in pieces) == n and [r for r, _ in pieces] == sorted(r for r, _ in pieces)
        assert all(r % align == 0 for r, _ in pieces)
        assert all(c * per < ops.DESC_SPAN or c == 1 for _, c in pieces)


def test_grouped_wide_gate_and_packed_columns():
    """Host logic of the rank-320 grouped projections (ops.grouped_wide_ok / ops.packed_columns, round 6): the layout lora.LoraBank
    gives a q | k | v trio -- A and Bup stacked, Bup^T stacked, A^T as column blocks of one [K, 3r] matrix -- is accepted, anything
    else stays on the per-site path; the attention backward's packed [dQ | dK | dV] is recognised in place."""
    r, C = 320, 640
    x = _x(2048, C)
    S16 = torch.empty(8, r, dtype=torch.bfloat16)
    a = torch.empty(3 * r, C, dtype=torch.bfloat16)
    b = torch.empty(3 * C, r, dtype=torch.bfloat16)
    bt = torch.empty(3 * r, C, dtype=torch.bfloat16)
    atc = torch.empty(C, 3 * r, dtype=torch.bfloat16)

    def sites(at_cols=True, rank=r):
        return [types.SimpleNamespace(rank=rank, a16=a[g * r:(g + 1) * r], b16=b[g * C:(g + 1) * C], bt16=bt[g * r:(g + 1) * r],
                                      at16=atc[:, g * r:(g + 1) * r] if at_cols else torch.empty(C, r, dtype=torch.bfloat16)) for g in range(3)]
    packs = [types.SimpleNamespace(N=C, K=C, bias=None) for _ in range(3)]
    assert ops.grouped_wide_ok(x, packs, sites(), S16)
    assert ops.grouped_wide_ok(x, packs[:2], sites()[:2], S16, need_dx=False)                  # the text k | v pair
    assert not ops.grouped_wide_ok(x, packs, sites(at_cols=False), S16)                         # A^T not laid out as column blocks
    assert ops.grouped_wide_ok(x, packs, sites(at_cols=False), S16, need_dx=False)
    assert not ops.grouped_wide_ok(x, packs, sites(rank=32), S16)                               # rank 32 has its own grouped kernel
    assert not ops.grouped_wide_ok(x, packs, list(reversed(sites())), S16)                      # not stacked in this order
    assert not ops.grouped_wide_ok(x, [types.SimpleNamespace(N=C, K=C, bias=torch.zeros(C))] + packs[1:], sites(), S16)
    assert not ops.grouped_wide_ok(x, [types.SimpleNamespace(N=480, K=C, bias=None)] * 3, sites(), S16)   # width not a multiple of 320
    assert not ops.grouped_wide_ok(x, packs, sites(), None)
    d = torch.empty(4, 512, 3 * C, dtype=torch.bfloat16)
    dq, dk, dv = (d[..., g * C:(g + 1) * C].reshape(2048, C) for g in range(3))
    cat = ops.packed_columns((dq, dk, dv), 2048, C)
    assert cat is not None and cat.data_ptr() == d.data_ptr() and cat.shape == (2048, 3 * C) and cat.stride() == (3 * C, 1)
    assert ops.packed_columns((dq, dv, dk), 2048, C) is None and ops.packed_columns((dq.contiguous(), dk, dv), 2048, C) is None
