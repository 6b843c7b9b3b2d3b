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
