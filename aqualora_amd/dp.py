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


def logged_loss(loss, group=None):
    """The loss value the reference LOGS: ``accelerator.gather(loss.repeat(train_batch_size)).mean()`` (train/ppft_train.py:1054),
    i.e. the mean of the per-rank batch-mean losses (every rank contributes the same number of copies).  Off the step's critical path:
    one scalar all-reduce, called only when a log line is written.  Returns a Python float."""
    t = loss.detach().float().reshape(1).clone()
    allreduce_mean_(t, group)
    return float(t.item())


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


class AqlComm:
    """RCCL communicator behind our own C-ABI (include/aqualora_hip.h: aql_comm_*), one per process / GPU.

    Unlike ProcessGroupNCCL the collectives are enqueued on the stream the CALLER names (torch's current stream by default),
    so the trainer can fork them onto a side stream under the rest of backward and capture them into the step's HIP graph
    (ppft.PPFTTrainer, "overlap" exchange).  The 128-byte id travels over the already initialised torch.distributed group --
    the launcher's rendezvous, plumbing -- or is created locally for a single-rank communicator (tests, AQL_FORCE_ALLREDUCE)."""

    def __init__(self, group=None, single=False):
        import ctypes
        from . import _lib as L
        self._L = L
        if single or not (dist.is_available() and dist.is_initialized()):
            self.rank, self.world = 0, 1
        else:
            self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        idbuf = ctypes.create_string_buffer(128)
        err = ""
        if self.rank == 0:
            try:
                L.call("aql_comm_unique_id", idbuf)
            except L.AqlError as e:   # must reach every rank: the others are about to wait in the broadcast below
                err = str(e)
        box = [bytes(idbuf.raw), err]
        if self.world > 1:
            src = dist.get_global_rank(group, 0) if group is not None else 0
            dist.broadcast_object_list(box, src=src, group=group)
        if box[1]:
            raise L.AqlError(f"aql_comm_unique_id failed on rank 0: {box[1]}")
        self._id = ctypes.create_string_buffer(box[0], 128)
        self.handle = ctypes.c_void_p()
        L.call("aql_comm_init", self._id, self.world, self.rank, ctypes.byref(self.handle))
        n = L.call_raw("aql_comm_size", self.handle)
        if n != self.world:
            raise L.AqlError(f"aql_comm_init: communicator reports {n} ranks, expected {self.world}")

    def _stream(self, stream):
        return self._L.c_p((stream or torch.cuda.current_stream()).cuda_stream)

    def all_reduce_(self, flat, average=True, stream=None):
        if flat.dtype != torch.float32 or not flat.is_contiguous():
            raise self._L.AqlError("AqlComm.all_reduce_: contiguous fp32 only")
        self._L.call("aql_comm_all_reduce_f32", self.handle, self._L.ptr(flat), flat.numel(), int(bool(average)),
                     self._stream(stream))
        return flat

    def reduce_scatter(self, send, recv, average=True, stream=None):
        if send.numel() != recv.numel() * self.world:
            raise self._L.AqlError("AqlComm.reduce_scatter: send must hold world x recv elements")
        self._L.call("aql_comm_reduce_scatter_f32", self.handle, self._L.ptr(send), self._L.ptr(recv), recv.numel(),
                     int(bool(average)), self._stream(stream))
        return recv

    def all_gather(self, send, recv, stream=None):
        nbytes = send.numel() * send.element_size()
        if recv.numel() * recv.element_size() != nbytes * self.world:
            raise self._L.AqlError("AqlComm.all_gather: recv must hold world x send bytes")
        self._L.call("aql_comm_all_gather", self.handle, self._L.ptr(send), self._L.ptr(recv), nbytes, self._stream(stream))
        return recv

    def broadcast_(self, t, root=0, stream=None):
        self._L.call("aql_comm_broadcast", self.handle, self._L.ptr(t), t.numel() * t.element_size(), int(root),
                     self._stream(stream))
        return t

    def self_test(self, timeout_s=60.0):
        """One captured, forked all-reduce replayed twice and checked, with a deadline: the trainer only switches to the captured
        exchange when this passes on the box it runs on (the captured form cannot be exercised across ranks on the 1-GPU
        development boxes; a communicator that fails or stalls here is aborted and the torch.distributed exchange is used)."""
        import time
        dev = torch.device("cuda", torch.cuda.current_device())
        buf = torch.ones(1 << 20, dtype=torch.float32, device=dev)   # 4 MB: beyond RCCL's single-channel / low-latency protocol sizes
        side = torch.cuda.Stream(device=dev)
        cap = torch.cuda.Stream(device=dev)
        cap.wait_stream(torch.cuda.current_stream())
        try:
            with torch.cuda.stream(cap):
                self.all_reduce_(buf, average=False)      # eager warm-up: RCCL sets its channels up outside the capture
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    buf.mul_(2.0)
                    side.wait_stream(torch.cuda.current_stream())
                    self.all_reduce_(buf, average=False, stream=side)
                    torch.cuda.current_stream().wait_stream(side)
                    buf.add_(1.0)
                g.replay()
                g.replay()
                done = torch.cuda.Event()
                done.record()
            t0 = time.time()
            while not done.query():
                if time.time() - t0 > timeout_s:
                    return False, f"captured all-reduce did not complete within {timeout_s:.0f} s"
                time.sleep(0.01)
            torch.cuda.current_stream().wait_stream(cap)
            w = float(self.world)
            want = (2.0 * w * w + 1.0) * 2.0 * w + 1.0    # eager: w; replay 1: 2w*w + 1; replay 2: (2w*w + 1) * 2w + 1
            got = buf.cpu()
            if not bool((got == want).all()):
                return False, f"captured all-reduce returned {float(got[0])}, expected {want}"
            self._keep = (g, buf, side, cap)
            return True, "ok"
        except Exception as e:   # noqa: BLE001  (any failure means: do not use the captured exchange)
            return False, f"{type(e).__name__}: {e}"

    def abort(self):
        if self.handle:
            self._L.call_raw("aql_comm_abort", self.handle)
            self.handle = None

    def destroy(self):
        if getattr(self, "handle", None):
            self._L.call_raw("aql_comm_destroy", self.handle)
            self.handle = None


def make_comm(group=None):
    """The communicator of the captured / overlapped exchange, or (None, reason) when it is not to be used: no exchange active,
    a non-RCCL backend (the gloo CPU tests), AQL_COMM=0, or a failed self-test.

    DEFAULT since round 6 (north_star: "RCCL all-reduce ... overlapped with backward"; the reference's DDP reducer fires from
    backward, ppft_train.py:905-912,1058): every data-parallel run takes the captured, hook-driven aql_comm_* exchange when the
    start-up self-test passes on ALL ranks -- RCCL loadable, ncclCommInitRank, communicator size == world size, and a captured,
    forked all-reduce of 4 MB replayed twice against its closed form under a deadline (AqlComm.self_test).  Any failure makes every
    rank fall back TOGETHER to the torch.distributed exchange, with the reason on stderr and in `comm_note` (bench.py prints it).
    AQL_COMM=0 selects the torch.distributed exchange outright."""
    if not exchange_active(group):
        return None, "no exchange (single rank)"
    if os.environ.get("AQL_COMM", "1") == "0":
        return None, "AQL_COMM=0 (torch.distributed exchange requested)"
    if dist.get_backend(group) != "nccl" or not torch.cuda.is_available():
        return None, f"backend {dist.get_backend(group)}"
    if group in _COMMS:                 # one communicator (and one self-test) per process group
        return _COMMS[group]
    _COMMS[group] = res = _make_comm(group)
    if res[0] is None:
        import sys
        print(f"[aqualora_amd.dp] rank {dist.get_rank(group)}: captured aql_comm_* exchange NOT in use -- {res[1]}; falling back to "
              "the torch.distributed exchange (not overlapped at rank 32)", file=sys.stderr, flush=True)
    return res


_COMMS = {}


def _agree(ok, group):
    """MIN over ranks of a local yes / no, through the launcher's torch.distributed group."""
    flag = torch.tensor([1.0 if ok else 0.0], device="cuda")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return flag.item() >= 1.0


def _make_comm(group):
    comm, ok, why = None, False, ""
    # (1) every rank must be able to bind RCCL BEFORE anyone enters the collective ncclCommInitRank: a rank whose dlopen / dlsym
    # fails would otherwise leave the others blocked inside it, out of reach of the agreement below.
    from . import _lib as L
    try:
        have = L.call_raw("aql_comm_available") == 1
        if not have:
            why = "RCCL not loadable: " + L.load().aql_last_error().decode("utf-8", "replace")
    except Exception as e:   # noqa: BLE001
        have, why = False, f"{type(e).__name__}: {e}"
    if not _agree(have, group):
        return None, f"RCCL unavailable ({why or 'on another rank'})"
    try:
        comm = AqlComm(group)
        ok, why = comm.self_test()
    except Exception as e:   # noqa: BLE001
        why = f"aql_comm_init failed: {e}"
    # every rank must take the same decision: a rank that fell back alone would wait in a torch.distributed collective forever
    if not _agree(ok, group):
        if comm is not None:
            try:
                comm.abort()     # also takes a captured all-reduce that the self-test's deadline left in flight down with it
            except Exception:   # noqa: BLE001
                pass
        return None, f"self-test failed ({why or 'on another rank'})"
    import atexit
    atexit.register(comm.destroy)    # ncclCommDestroy at interpreter exit (a trainer may also call comm.destroy() itself)
    return comm, "ok"


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
    forward pass, so that the replicas stay identical.  The ~200 BatchNorm buffers of the SecretDecoder travel as ONE collective
    per dtype (fp32 statistics; int64 `num_batches_tracked` counters): packed into a flat tensor, broadcast, scattered back with
    one multi-tensor copy -- not one ~10 us collective per buffer."""
    if world_size(group) <= 1:
        return
    by_dtype = {}
    for b in module.buffers():
        if b.numel():
            by_dtype.setdefault(b.dtype, []).append(b)
    root = dist.get_global_rank(group, src) if group is not None else src
    for bufs in by_dtype.values():
        flat = torch.cat([b.detach().reshape(-1) for b in bufs])
        dist.broadcast(flat, src=root, group=group)
        outs, off = [], 0
        for b in bufs:
            outs.append(flat[off:off + b.numel()].view_as(b))
            off += b.numel()
        with torch.no_grad():
            try:
                torch._foreach_copy_(bufs, outs)
            except (AttributeError, RuntimeError):
                for b, o in zip(bufs, outs):
                    b.copy_(o)


class ModuleGradExchange:
    """DDP's grad-ready buckets for an ordinary trainable module (the SecretDecoder of rob_enhance_finetune.py:917-919,1037,
    ~26 MB of fp32 gradients), overlapped with its backward pass.

    Every parameter's ``.grad`` is a view into ONE flat fp32 buffer laid out in reverse parameter order -- the order in which
    backward produces the gradients (classifier first, stem last) -- cut into ``n_buckets`` contiguous ranges at parameter
    boundaries (for EfficientNet-B1 with 4 buckets: head + the last MBConv stages | ... | stem + the first stages).  A
    post-accumulate-grad hook on every parameter counts its bucket down; the hook that completes a bucket hands its range to the
    collective at once: ``aql_comm_all_reduce_f32`` on a forked side stream when the trainer has our RCCL communicator (``comm``),
    else torch.distributed's asynchronous all-reduce -- either way it runs under the rest of backward.  ``finish()`` joins.
    No ``torch.cat`` pack, no copy-back: the optimizer reads the averaged gradients where the collective left them.
    ``zero_grad()`` zeroes the flat buffer and keeps the views (``optimizer.zero_grad()`` would drop them: set_to_none)."""

    def __init__(self, module, group=None, comm=None, n_buckets=4):
        self.group, self.comm = group, comm
        self.params = [p for p in reversed(list(module.parameters())) if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.offsets, off = [], 0
        for p in self.params:
            if p.dtype != torch.float32:
                raise ValueError("ModuleGradExchange: fp32 parameters only")
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            self.offsets.append(off)
            off += p.numel()
        # buckets: contiguous runs of parameters of about n / n_buckets elements each
        self.bucket_of, self.ranges = [], []
        target, lo, b = max(1, n // max(1, n_buckets)), 0, 0
        for i, p in enumerate(self.params):
            self.bucket_of.append(b)
            end = self.offsets[i] + p.numel()
            if end - lo >= target and b < n_buckets - 1 and i + 1 < len(self.params):
                self.ranges.append((lo, end))
                lo, b = end, b + 1
        self.ranges.append((lo, n))
        self.sizes = [sum(1 for q in self.bucket_of if q == k) for k in range(len(self.ranges))]
        self.left = list(self.sizes)
        self.reducer = BucketedAllreduce(group)
        self.side = torch.cuda.Stream(device=dev) if (comm is not None and dev.type == "cuda") else None
        self.launched = []
        self._handles = [p.register_post_accumulate_grad_hook(lambda _p, _i=i: self._ready(_i)) for i, p in enumerate(self.params)]

    def _ready(self, i):
        # the hooks reduce the FLAT buffer: a parameter whose .grad is no longer a view of it (optimizer.zero_grad(set_to_none=True)
        # or module.zero_grad() between prepare and the step) would be stepped on its local, un-averaged gradient while the
        # collective averages zeros -- replicas would diverge silently.  Catch it here: copy the stray gradient in and re-attach.
        p, off = self.params[i], self.offsets[i]
        if p.grad is not None and p.grad.data_ptr() != self.flat.data_ptr() + 4 * off:
            view = self.flat[off:off + p.numel()].view_as(p)
            view.copy_(p.grad)
            p.grad = view
        k = self.bucket_of[i]
        self.left[k] -= 1
        if self.left[k] == 0:
            self._launch(k)

    def _launch(self, k):
        lo, hi = self.ranges[k]
        self.launched.append(k)
        if not exchange_active(self.group):
            return
        if self.comm is not None:
            main = torch.cuda.current_stream()
            self.side.wait_stream(main)            # fork behind the kernels that produced this bucket's gradients
            with torch.cuda.stream(self.side):
                self.comm.all_reduce_(self.flat[lo:hi], average=True)
        else:
            self.reducer.launch(self.flat[lo:hi])

    def finish(self):
        """After backward: every bucket was launched from a hook (a parameter that received no gradient leaves its bucket
        open: it is launched here); wait for the collectives."""
        for k in range(len(self.ranges)):      # ascending bucket index on every rank: the collective order cannot differ across ranks
            if self.left[k] != 0 and k not in self.launched:
                self._launch(k)
        if self.comm is not None and self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)
        self.reducer.finish()
        order = list(self.launched)
        self.left = list(self.sizes)
        self.launched = []
        return order

    def zero_grad(self):
        self.flat.zero_()
        for p, off in zip(self.params, self.offsets):      # re-attach views an optimizer.zero_grad(set_to_none=True) dropped
            if p.grad is None or p.grad.data_ptr() != self.flat.data_ptr() + 4 * off:
                p.grad = self.flat[off:off + p.numel()].view_as(p)

    def close(self):
        for h in self._handles:
            h.remove()
        self._handles = []
