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


def exchange_active(group=None):
    """True when a gradient collective has to run: more than one rank, or AQL_FORCE_ALLREDUCE=1 (exercises the RCCL
    path on a single GPU)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return world_size(group) > 1 or bool(os.environ.get("AQL_FORCE_ALLREDUCE"))


def bucket_count(nbytes):
    """Buckets for the flat LoRA gradient buffer; 1 = no bucketing (one grouped weight-gradient launch inside the
    backward graph, one collective after it).

    Measured on MI355X (bench.py under torchrun, collectives forced on one rank): every extra graph boundary costs
    ~0.35 ms of idle GPU, so at r=32 (54 MB, ~0.5 ms on the wire over 8 GPUs) splitting cannot win and the exchange
    stays ONE collective; at r=320 (543 MB, ~5 ms on the wire) eight 68 MB buckets cost 1.9 ms of boundaries on a
    73 ms step and hide all but the last bucket's transfer under the weight-gradient GEMMs.  xGMI is point-to-point
    (a ring is bound by one ~153 GB/s link and pays its latency per collective), so buckets stay >= 64 MB.
    AQL_BUCKETS overrides (tests)."""
    if os.environ.get("AQL_BUCKETS"):
        return max(1, int(os.environ["AQL_BUCKETS"]))
    if nbytes < (128 << 20):
        return 1
    return int(min(8, nbytes // (64 << 20)))


class BucketedAllreduce:
    """Asynchronous mean-all-reduce of slices of one flat buffer, one collective per bucket.

    ``launch(chunk)`` enqueues the collective behind everything already queued on the CURRENT stream (ProcessGroupNCCL
    records an event there and runs the collective on its own stream), so kernels launched afterwards on the current
    stream -- the next bucket's weight-gradient GEMMs -- overlap with it.  ``finish()`` makes the current stream wait
    for all of them (and applies the 1/world factor on backends without ReduceOp.AVG)."""

    def __init__(self, group=None):
        self.group = group
        self.pending = []

    def launch(self, chunk):
        if not exchange_active(self.group) or chunk.numel() == 0:
            return
        if dist.get_backend(self.group) == "nccl":
            w = dist.all_reduce(chunk, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
            self.pending.append((w, None))
        else:
            w = dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self.pending.append((w, chunk))

    def finish(self):
        n = world_size(self.group)
        for w, chunk in self.pending:
            w.wait()
            if chunk is not None:
                chunk.div_(n)
        self.pending = []


def allreduce_module_grads_(params, group=None, bucket_bytes=64 << 20):
    """DDP's gradient averaging for an ordinary module (the SecretDecoder in rob_enhance_finetune.py, ~26 MB of fp32
    gradients): gradients are packed into flat buckets of at most ``bucket_bytes`` (reverse parameter order, the order in
    which backward produces them), each bucket is mean-all-reduced asynchronously, then scattered back."""
    if not exchange_active(group):
        return
    grads = [p.grad for p in reversed(list(params)) if p.grad is not None]
    if not grads:
        return
    red = BucketedAllreduce(group)
    buckets, cur, size = [], [], 0
    for g in grads:
        nb = g.numel() * g.element_size()
        if cur and size + nb > bucket_bytes:
            buckets.append(cur)
            cur, size = [], 0
        cur.append(g)
        size += nb
    buckets.append(cur)
    flats = []
    for b in buckets:
        flat = torch.cat([g.reshape(-1) for g in b])
        red.launch(flat)
        flats.append(flat)
    red.finish()
    for b, flat in zip(buckets, flats):
        off = 0
        for g in b:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()


def broadcast_(tensor, group=None, src=0):
    """DDP's construction-time parameter sync (what accelerate's ``prepare`` does when it wraps the trainable modules,
    train/ppft_train.py:905-912): rank ``src``'s values overwrite every rank's copy.  The reference runs with
    ``seed=None`` by default, so without this every rank would start from its own LoRA / mapper draw and the replicas
    would drift apart silently while averaging gradients.  No-op on a single rank."""
    if world_size(group) <= 1:
        return tensor
    dist.broadcast(tensor, src=dist.get_global_rank(group, src) if group is not None else src, group=group)
    return tensor


def broadcast_module_(module, group=None, src=0):
    """Parameters and buffers of an ordinary module (the SecretDecoder of rob-finetune) from rank ``src``."""
    if world_size(group) <= 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        if t.numel():
            broadcast_(t.data, group, src)


def broadcast_buffers_(module, group=None, src=0):
    """DDP(broadcast_buffers=True): rank ``src``'s buffers (BatchNorm running statistics) overwrite everyone's before a
    forward pass, so that the replicas stay identical."""
    if world_size(group) <= 1:
        return
    for b in module.buffers():
        if b.numel():
            dist.broadcast(b, src=src, group=group)
