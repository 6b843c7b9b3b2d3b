"""Data-parallel gradient exchange for the PPFT step (reference: DDP's all-reduce(mean) fired from
``accelerator.backward``, train/ppft_train.py:1058).

One process per GPU; the LoRA + mapper gradients already live in ONE flat fp32 buffer laid out in gradient-ready
order (lora.LoraBank), so the exchange is one collective (or a few large buckets) over RCCL/xGMI instead of DDP's
many 25 MB buckets.  Backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests.
"""
import os

import torch
import torch.distributed as dist


def world_size(group=None):
    return dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1


def allreduce_mean_(flat, group=None, bucket_elems=None):
    """In-place mean over ranks of a flat tensor, optionally in buckets of ``bucket_elems`` (front buckets first:
    with the bank's reverse-traversal layout they are complete first in backward)."""
    w = world_size(group)
    if w == 1 and not (dist.is_available() and dist.is_initialized() and os.environ.get("AQL_FORCE_ALLREDUCE")):
        return flat  # (AQL_FORCE_ALLREDUCE=1 exercises the collective on a single GPU)
    backend = dist.get_backend(group)
    n = flat.numel()
    step = n if not bucket_elems else int(bucket_elems)
    for lo in range(0, n, step):
        chunk = flat[lo:min(n, lo + step)]
        if backend == "nccl":
            dist.all_reduce(chunk, op=dist.ReduceOp.AVG, group=group)
        else:
            dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=group)
            chunk.div_(w)
    return flat
