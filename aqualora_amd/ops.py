"""torch.autograd.Function wrappers over the gfx950 C-ABI kernels.

Conventions
  * activations are bf16; 4-D tensors are logical NCHW with ``channels_last`` strides (physically NHWC), so a
    feature map [B,C,H,W] and its token view [B,H*W,C] share memory and no permute kernel ever runs;
  * frozen base weights are pre-packed once per host module (``pack_linear`` / ``pack_conv3x3``);
  * LoRA weight gradients are accumulated straight into the flat fp32 gradient buffer owned by
    ``aqualora_amd.lora.LoraBank`` (the Function returns ``None`` for them), the gradient of the per-message
    diagonal ``S`` is returned through autograd so that it reaches MapperNet.
Every function here requires a CUDA(HIP) tensor and the built shared library: there is no fallback.
"""
import os

import torch

from . import _lib as L

CL = torch.channels_last
_WS = {}
_WS_RETIRED = []
_GN = {}


def _skey(device):
    # one workspace per (device, stream): kernels of concurrent streams must not share split-K slabs / GN scratch
    return (device.index if device.index is not None else torch.cuda.current_device(),
            torch.cuda.current_stream(device).cuda_stream)


# ---- a split-K convolution whose finalize launch was held back for the GroupNorm behind it (aql_conv3x3_*_defer, round 6)
# OFF by default (AQL_DEFER_FINALIZE=1 enables it): measured round 6 on MI355X -- config 2 step 21.13 -> 21.45 ms, config 3 46.27 -> 46.45,
# config 4 264.7 -> 264.3 ms per image (profiles/r06_defer_finalize.txt).  A kernel boundary inside the captured graph costs ~1.5 us
# (finalize + GroupNorm back to back: 7.3-12 us, GroupNorm alone 4.1-5.6), while the GroupNorm workgroup -- 20-40 channels of every pixel of
# ONE (sample, group) -- reads the fp32 slabs as 80-160-byte segments at twice the bytes: 8.1-13 us in one launch on 8 x 8 / 16 x 16 maps,
# 23-58 us against 15-37 on 32 x 32 maps.  The launch count was never the cost.  Kept as an entry point with its bit-identity test.
DEFER_FINALIZE = os.environ.get("AQL_DEFER_FINALIZE", "0") == "1"
_PENDING = {}     # (device, stream) -> _Pending: at most one per slab buffer (defer_workspace)


class _Pending:
    """`t` (a dense bf16 [M, N] map, channels-last) is still `splits` fp32 partial slabs in the stream's workspace.  Whoever touches
    the workspace or reads `t` next either consumes the slabs itself (GroupNormSiluFn forward / backward: the finalize is the first
    pass of the GroupNorm kernel) or runs the finalize launch first (`flush_pending`): nothing ever reads an unfinished tensor."""
    __slots__ = ("t", "ws", "splits", "M", "N", "bias", "rowbias", "rowbias_ld", "rps", "residual")

    def finalize(self):
        L.call("aql_splitk_finalize", L.ptr(self.ws), self.splits, self.M, self.N, L.ptr(self.bias), L.ptr(self.rowbias),
               int(self.rowbias_ld), int(self.rps), L.ptr(self.residual), self.N, L.ptr(self.t), self.N, L.stream_ptr())


def flush_pending(device):
    p = _PENDING.pop(_skey(device), None)
    if p is not None:
        p.finalize()


def take_pending(t):
    """The pending record of tensor `t` (matched by storage: twin views share their buffer's), removed from the table; else None."""
    key = _skey(t.device)
    p = _PENDING.get(key)
    if p is not None and p.t.untyped_storage().data_ptr() == t.untyped_storage().data_ptr():
        return _PENDING.pop(key)
    return None


_WS_DEFER = {}


def defer_workspace(device, nfloats=64 << 20):   # (the size of the shared workspace: the split count is chosen against it)
    """The slab buffer of the held-back launches, per (device, stream), apart from the shared split-K workspace: launches between a
    deferred convolution and its GroupNorm (the shortcut GEMM's backward runs between conv2's and norm2's) do not touch it.  Asking
    for it finishes whatever it still holds."""
    flush_pending(device)
    key = _skey(device)
    ws = _WS_DEFER.get(key)
    if ws is None or ws.numel() < nfloats:
        if ws is not None:
            _WS_RETIRED.append(ws)
        ws = torch.empty(nfloats, dtype=torch.float32, device=device)
        _WS_DEFER[key] = ws
    return ws


def flush_all_pending():
    """End of a pass: nothing may stay unfinished (by construction a GroupNorm follows every deferred launch: this is the net)."""
    for key in list(_PENDING):
        _PENDING.pop(key).finalize()


def workspace(device, nfloats=64 << 20):
    key = _skey(device)
    ws = _WS.get(key)
    if ws is None or ws.numel() < nfloats:
        if ws is not None:
            _WS_RETIRED.append(ws)   # captured graphs may have baked the old pointer in: never hand it back to the allocator
        ws = torch.empty(nfloats, dtype=torch.float32, device=device)
        _WS[key] = ws
    return ws


def _gn_scratch(device, B):
    key = _skey(device)
    t = _GN.get(key)
    need = B * 128 * 32 * 2 + B * 32 * 2
    if t is None or t.numel() < need:
        t = torch.empty(max(need, 1 << 17), dtype=torch.float32, device=device)
        _GN[key] = t
    return t


# ---------------------------------------------------------------------------------- branch-level concurrency
BRANCH_STREAMS = 0   # >0: independent sibling ops (attention q/k/v projections) are issued on this many side streams.
# EXPERIMENTAL, eager only: nested stream forks inside a hipGraph capture crash hipStreamEndCapture on this ROCm
# (measured round 1), so the captured trainer keeps it at 0.
_BR = {}


def parallel(fns):
    """Run independent thunks; with BRANCH_STREAMS > 0 all but the first go to side HIP streams that fork from and
    join back into the current stream (autograd replays each branch's backward on its own stream too)."""
    if BRANCH_STREAMS <= 0 or len(fns) < 2:
        return [f() for f in fns]
    main = torch.cuda.current_stream()
    key = (main.device.index, main.cuda_stream)
    pool = _BR.get(key)
    if pool is None or len(pool) < BRANCH_STREAMS:
        pool = [torch.cuda.Stream(device=main.device) for _ in range(BRANCH_STREAMS)]
        _BR[key] = pool
    outs = [None] * len(fns)
    used = []
    for i, f in enumerate(fns[1:], start=1):
        st = pool[(i - 1) % len(pool)]
        st.wait_stream(main)
        with torch.cuda.stream(st):
            outs[i] = f()
        used.append(st)
    outs[0] = fns[0]()
    for st in used:
        main.wait_stream(st)
    if not torch.cuda.is_current_stream_capturing():
        for o in outs[1:]:
            if torch.is_tensor(o):
                o.record_stream(main)
    return outs


def join_branches():
    """Make the current stream wait for every branch stream forked from it (and from any other stream): needed
    before work that consumes branch results outside autograd's own synchronisation (deferred dW/dS) and before a
    graph capture ends."""
    cur = torch.cuda.current_stream()
    for pool in _BR.values():
        for st in pool:
            cur.wait_stream(st)


def _req(t, name):
    if not t.is_cuda:
        raise L.AqlError(f"{name}: the HIP path needs a GPU tensor (got {t.device}); there is no CPU fallback")
    if _PENDING:
        p = take_pending(t)
        if p is not None:      # an op other than the GroupNorm reads a tensor whose finalize was held back: finish it now
            p.finalize()
    return t


# ------------------------------------------------------------------------------ twin ("dual") batches
# The PPFT step runs the frozen U-Net twice per sample: the "clean" pass (no LoRA, no gradient, ppft_train.py:1026-1029)
# and the watermarked pass (LoRA, gradient, :1032-1035).  At batch 4 every kernel of either pass is latency- or
# weight-streaming-bound, so the two passes are run as ONE launch per op over a batch of 2B: clean samples in the first
# half (their scale rows are all-zero, which is exactly the reference's clean pass), watermarked samples in the second.
# Autograd only ever sees the second half: every activation tensor is the [B:] view of a 2B-sample buffer whose storage
# is registered here, each op's forward finds the full buffer behind its input view (`_full`), launches on all 2B samples
# and returns the [B:] view of its (registered) output.  Backward functions are unchanged -- they work on the saved
# half-views.  Measured on MI355X (tools/exp_batched_fwd.py): forward of both passes 15.2 ms on two streams -> 13.9 ms.
class _Dual:
    def __init__(self):
        self.storages = {}   # storage data_ptr -> the full tensor (keeps it alive for the duration of the forward)

    def register(self, full):
        self.storages[full.untyped_storage().data_ptr()] = full


DUAL = None
_TWIN_SKIP = True   # skip the LoRA branch on clean tiles (module attribute: probes flip it; the environment hook left in round 6)


def dual_begin():
    global DUAL
    DUAL = _Dual()
    return DUAL


def dual_end():
    global DUAL
    DUAL = None
    _CPAD.clear()


# Channel-padded inputs: a [B, C, H, W] view of a channels-last buffer that is Cp > C channels wide with the channels C..Cp-1 ZERO
# (written so by ppft's prologue kernel at conv_in's packed width).  conv3x3 then reads the buffer as it is instead of building the
# padded copy (a fill + a strided copy in front of the first conv of every forward).  Keyed by (address, shape, strides); cleared with
# the twin registry.
_CPAD = set()


def register_cpad(x):
    _CPAD.add((x.data_ptr(), tuple(x.shape), tuple(x.stride())))
    return x


def is_cpad(x, cp):
    return x.dim() == 4 and x.stride(1) == 1 and x.stride(3) == cp and (x.data_ptr(), tuple(x.shape), tuple(x.stride())) in _CPAD


def _twin_geometry(x):
    """(registered full tensor, elements per half, dim-0 stride) of a second-half view, or None.  The batch is the outermost
    dimension of every activation layout used here, so the halves are `half` elements apart whatever the view's shape;
    a size-1 dim 0 (batch 1) carries no usable stride of its own (torch normalises it in view ops)."""
    if DUAL is None or x is None:
        return None
    full = DUAL.storages.get(x.untyped_storage().data_ptr())
    if full is None:
        return None
    half = full.numel() // 2
    if x.storage_offset() < half:
        raise L.AqlError("twin batch: a view that is not inside the second half of its buffer reached a kernel")
    return full, half, (x.stride(0) if x.shape[0] > 1 else half)


def _full(x):
    """x = a view inside the second (batch-major) half of a registered twin buffer -> the view of BOTH halves (same inner
    strides, 2x dim 0); None when x is an ordinary tensor or twin mode is off."""
    geo = _twin_geometry(x)
    if geo is None:
        return None
    _, half, s0 = geo
    return x.detach().as_strided((2 * x.shape[0],) + tuple(x.shape[1:]), (s0,) + tuple(x.stride()[1:]),
                                 x.storage_offset() - half)


def clean_twin(x):
    """The clean-pass counterpart (same view of the FIRST half of the twin buffer) of a second-half view, detached.
    Call while the twin registry is alive (before dual_end)."""
    geo = _twin_geometry(x)
    if geo is None:
        raise L.AqlError("clean_twin: not a twin tensor")
    return x.detach().as_strided(tuple(x.shape), x.stride(), x.storage_offset() - geo[1])


def make_twin(first, second):
    """Build a registered twin buffer from two equally shaped tensors; returns the second-half view (what autograd sees)."""
    full = torch.cat([first, second], dim=0)
    if first.dim() == 4:
        full = full.contiguous(memory_format=CL)
    DUAL.register(full)
    return full[first.shape[0]:]


def _alloc(shape, dtype, device, twin, cl=False):
    """-> (tensor the kernel writes, tensor autograd sees).  twin: allocate 2x dim 0, register, hand out the [n:] view."""
    if not twin:
        t = torch.empty(shape, dtype=dtype, device=device, memory_format=CL) if cl else torch.empty(shape, dtype=dtype, device=device)
        return t, t
    shp = (2 * shape[0],) + tuple(shape[1:])
    full = torch.empty(shp, dtype=dtype, device=device, memory_format=CL) if cl else torch.empty(shp, dtype=dtype, device=device)
    DUAL.register(full)
    return full, full[shape[0]:]


def _need_full(t, what):
    f = _full(t)
    if f is None:
        raise L.AqlError(f"twin batch: {what} is not a twin tensor although the activation is")
    return f


def as_cl(x):
    """Logical NCHW -> channels_last storage (no-op when already so)."""
    return x.contiguous(memory_format=CL)


def nhwc_view(x):
    return x.permute(0, 2, 3, 1)


# --------------------------------------------------------------------------------------------- packing
_PACK_SERIAL = [0]


def next_pack_serial():
    """Every packed (kernel-layout) weight copy gets a fresh number: a captured sampling loop names the copies it was captured
    on (inference._weights_key) and is dropped when any of them has been replaced (fuse_lora, re-initialised weights)."""
    _PACK_SERIAL[0] += 1
    return _PACK_SERIAL[0]


class PackedLinear:
    """bf16 kernel-layout copies of a frozen nn.Linear / 1x1 conv: W [N,K], W^T [K,N], bias [N]."""

    def __init__(self, weight, bias):
        self.serial = next_pack_serial()
        w = weight.detach().reshape(weight.shape[0], -1)
        self.N, self.K = w.shape
        self.w = w.to(torch.bfloat16).contiguous()
        self.wt = self.w.t().contiguous()
        self.bias = None if bias is None else bias.detach().to(torch.bfloat16).contiguous()


class PackedConv3x3:
    """Wk [Cout, 9*Cin] (tap-major, channel-minor) for forward, Wt [Cin, 9*Cout] for backward-data."""

    def __init__(self, weight, bias, stride):
        self.serial = next_pack_serial()
        Cout, Cin = weight.shape[0], weight.shape[1]
        self.Cin_real = Cin
        w = weight.detach().to(torch.bfloat16)
        if Cin % 8 != 0:  # conv_in: pad input channels with zeros
            pad = 8 - Cin % 8
            w = torch.cat([w, w.new_zeros(Cout, pad, 3, 3)], dim=1)
            Cin += pad
        self.Cout_real = Cout
        if Cout % 8 != 0:  # conv_out: pad output channels with zeros
            pad = 8 - Cout % 8
            w = torch.cat([w, w.new_zeros(pad, Cin, 3, 3)], dim=0)
            bias = None if bias is None else torch.cat([bias.detach(), bias.new_zeros(pad)])
            Cout += pad
        self.Cin, self.Cout, self.stride = Cin, Cout, int(stride)
        self.wk = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous()
        self.wt = w.permute(1, 2, 3, 0).reshape(Cin, 9 * Cout).contiguous()
        self.bias = None if bias is None else bias.detach().to(torch.bfloat16).contiguous()


# ------------------------------------------------------------------------------------- low-level calls
DESC_SPAN = 1 << 30   # the kernels' buffer descriptors cover 1 GiB from an operand's base (csrc/aql_gemm.cuh BUF_BYTES): larger
# activations are processed in row / sample chunks, each with its own descriptor base.  The C entry points refuse larger spans.


def span_chunks(n, bytes_per_unit, align=1):
    """[(first, count)] covering n units (rows / samples) so that each chunk's operand span stays below DESC_SPAN."""
    per = max(1, (DESC_SPAN - 1) // max(1, bytes_per_unit))
    if per >= n:
        return [(0, n)]
    per = max(align, per // align * align)
    return [(i, min(per, n - i)) for i in range(0, n, per)]


def gemm_bf16(A, B, bias=None, A2=None, B2=None, rowbias=None, rps=1, residual=None, out=None, lora_row0=0):
    """C[M,N] = A[M,K].B[N,K]^T (+A2.B2^T) + bias + rowbias[m//rps] + residual, bf16 with fp32 accumulation.
    lora_row0: rows below it have no A2.B2^T term (the clean half of a twin batch)."""
    M, K = A.shape
    N = B.shape[0]
    C = out if out is not None else torch.empty(M, N, dtype=torch.bfloat16, device=A.device)
    ws = workspace(A.device)
    ld = max(A.stride(0), C.stride(0), 0 if A2 is None else A2.stride(0), 0 if residual is None else residual.stride(0))
    align = rps if rowbias is not None else 1
    if align * ld * 2 >= DESC_SPAN:
        raise L.AqlError(f"gemm_bf16: one sample ({align} rows of {ld} elements) spans more than the 1 GiB an operand descriptor "
                         "covers; a row bias needs whole samples per launch (unsupported shape)")
    chunks = span_chunks(M, ld * 2, align)
    for r0, n in chunks:
        sl = slice(r0, r0 + n)
        Ac, Cc = A[sl], C[sl]
        A2c = None if A2 is None else A2[sl]
        rc = None if residual is None else residual[sl]
        rb = None if rowbias is None else rowbias[r0 // rps:]
        L.call("aql_gemm_bf16_ex", L.ptr(Ac), A.stride(0), L.ptr(B), B.stride(0), n, N, K,
               L.ptr(A2c), 0 if A2 is None else A2.stride(0), L.ptr(B2), 0 if B2 is None else B2.stride(0),
               0 if A2 is None else A2.shape[1], L.ptr(bias), L.ptr(rb), rps,
               L.ptr(rc), 0 if residual is None else residual.stride(0), L.ptr(Cc), C.stride(0), max(0, int(lora_row0) - r0),
               L.ptr(ws), ws.numel() * 4, L.stream_ptr())
    return C


_TN_SCRATCH = {}


def _tn_scratch(nelem, device):
    """per-(device, stream) bf16 scratch for the transposed operands of the wide-rank weight-gradient GEMMs"""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    buf = _TN_SCRATCH.get(key)
    if buf is None or buf.numel() < nelem:
        buf = torch.empty(max(nelem, 1 << 24), dtype=torch.bfloat16, device=device)
        _TN_SCRATCH[key] = buf
    return buf


def gemm_tn_acc(U, V, C, alpha=1.0):
    """C[P,Q] (fp32) += alpha * U[M,P]^T V[M,Q].  Narrow problems (a LoRA rank <= 32 side) use the transposing-loader
    kernel (normally through the grouped launch); wide-rank problems run on aql_gemm_tn_tr_f32: 128x128 tiles whose
    fragments are gathered from the row-major operands with LDS transpose reads (no transposed copies in HBM).
    AQL_TN_OLD=1 selects the previous wide path (transpose both operands once + pipelined NT kernels) for comparison."""
    M, P = U.shape
    Q = V.shape[1]
    if U.stride(1) == 1 and V.stride(1) == 1 and not TN_OLD:
        L.call("aql_gemm_tn_tr_f32", L.ptr(U), U.stride(0), L.ptr(V), V.stride(0), M, P, Q, float(alpha), L.ptr(C),
               C.stride(0), L.stream_ptr())
        return
    wide = (min(P, Q) > 32 and M % 8 == 0 and M * P * Q >= 1.5e9 and U.stride(1) == 1 and V.stride(1) == 1)  # >= 3 GFLOP
    if not wide:
        L.call("aql_gemm_tn_f32", L.ptr(U), U.stride(0), L.ptr(V), V.stride(0), M, P, Q, float(alpha), L.ptr(C),
               C.stride(0), L.stream_ptr())
        return
    st = L.stream_ptr()
    buf = _tn_scratch(M * (P + Q), U.device)
    Ut, Vt = buf[:M * P], buf[M * P:M * (P + Q)]
    L.call("aql_transpose_bf16", L.ptr(U), M, P, U.stride(0), L.ptr(Ut), st)
    L.call("aql_transpose_bf16", L.ptr(V), M, Q, V.stride(0), L.ptr(Vt), st)
    ws = workspace(U.device)
    L.call("aql_gemm_nt_f32_accum", L.ptr(Ut), M, L.ptr(Vt), M, P, Q, M, float(alpha), L.ptr(C), C.stride(0), L.ptr(ws),
           ws.numel() * 4, st)


# ------------------------------------------------------------------------ deferred (grouped) weight gradients
def plan_buckets(offsets, sizes, nbuckets):
    """Cut work items into <= nbuckets groups of CONTIGUOUS ranges of a flat buffer, balanced by size.

    offsets/sizes: start and length (elements) of the output region of each item inside the flat gradient buffer.
    Returns [(lo, hi, [item indices])...] ordered by lo; every item lies inside exactly one [lo, hi) and the ranges
    tile [min offset, max end) without gaps, so one collective per range exchanges exactly that group's results
    (plus any untouched elements between them, which are zeros on every rank).  Pure host logic (CPU-tested)."""
    order = sorted(range(len(offsets)), key=lambda i: offsets[i])
    if not order:
        return []
    total = sum(sizes)
    target = total / max(1, nbuckets)
    out, cur, acc, done = [], [], 0, 0
    for i in order:
        cur.append(i)
        acc += sizes[i]
        if acc >= target and len(out) < nbuckets - 1 and done + acc < total:
            out.append(cur)
            done += acc
            cur, acc = [], 0
    if cur:
        out.append(cur)
    res = []
    lo = offsets[order[0]]
    for k, items in enumerate(out):
        hi = max(offsets[i] + sizes[i] for i in items)
        if k + 1 < len(out):
            nxt = min(offsets[i] for i in out[k + 1])
            if nxt < hi:
                raise ValueError("plan_buckets: output regions overlap")
            hi = nxt
        res.append((lo, hi, items))
        lo = hi
    return res


class DeferredDW:
    """Collects the LoRA weight-gradient GEMMs (dA, dBup) and the dS reductions of a whole backward pass and runs
    them as ONE grouped launch each (`flush`).  They are off the backward critical path -- nothing consumes dA/dB/dS
    before the optimizer -- and as 384+192 separate small launches they were the largest single item of the step.
    Descriptors are written into pinned host tables and copied with one async H2D each (graph-capturable: the device
    addresses they hold are the capture pool's, identical on every replay).  Two tables: narrow problems (a LoRA rank
    <= 32 side: 128x32 tiles, `aql_gemm_tn_grouped`) and wide ones (rank > 32: 128x128 transpose-read tiles,
    `aql_gemm_tn_tr_grouped`); a problem neither kernel takes is launched on its own.

    Data-parallel runs use the bucketed form instead (`plan` / `run_bucket`): the problems are sorted by where their
    output lives in the flat gradient buffer and cut into a few contiguous buckets; the trainer launches bucket k and
    immediately hands its range to RCCL on the collective stream, so the all-reduce of bucket k runs under the
    weight-gradient GEMMs of bucket k+1 (the role DDP's grad-ready hooks play at ppft_train.py:1058)."""

    # kind -> (descriptor bytes, byte offset of first_block inside the descriptor, fill entry, grouped launch entry)
    KINDS = {"n": (80, 64, "aql_tn_desc_fill", "aql_gemm_tn_grouped_range"),
             "w": (96, 88, "aql_tntr_desc_fill", "aql_gemm_tn_tr_grouped"),   # TnTrDesc: 88-byte TnArgs, first_block
             "x": (96, 88, "aql_tntr160_desc_fill", "aql_gemm_tn_tr160_grouped")}   # round 6: 128 x 160 tiles for a side of 320 / 960 (rank 320)
    DS_BYTES = 48

    def __init__(self, device, max_sites=1024, defer_wide=True):
        self.device = device
        self.defer_wide = defer_wide   # kept for callers; wide problems are always grouped now
        self.allow_post = False        # set by a trainer whose exchange form runs `flush` (single GPU): enables the weight-side LoRA form
        self.host, self.dev = {}, {}
        for k, (nbytes, _, _, _) in self.KINDS.items():
            self.host[k] = torch.zeros(2 * max_sites * nbytes, dtype=torch.uint8).pin_memory()
            self.dev[k] = torch.zeros_like(self.host[k], device=device)
        self.ds_host = torch.zeros(max_sites * self.DS_BYTES, dtype=torch.uint8).pin_memory()
        self.ds_dev = torch.zeros_like(self.ds_host, device=device)
        self._uploads = []   # events recorded behind the async H2D copies of the tables (eager mode only)
        self.reset()

    def reset(self):
        self.n = {k: 0 for k in self.KINDS}      # descriptors per table
        self.blk = {k: 0 for k in self.KINDS}    # workgroups per table
        self.n_ds = self.blk_ds = 0
        self.keep = []
        self.items = []    # (C, kind, nblk, direct args or None) in arrival order; kind "d" = launched on its own
        self.buckets = None
        self.post = []     # weight-side LoRA sites: callables run between two grouped launches of `flush` (see wside_backward)

    @property
    def n_tn(self):
        return self.n["n"] + self.n["w"]

    def _host_tables_free(self):
        """Pinned memory does not protect an in-flight async H2D copy from later host writes: before the FIRST descriptor
        of a new step is written, wait until the previous step's table uploads have executed (the CPU can run a whole
        step ahead of a GPU-bound stream).  No-op while capturing (a captured memcpy node reads the table at every
        replay: a capture owns its DeferredDW, see PPFTTrainer.capture)."""
        if self._uploads:
            for ev in self._uploads:
                ev.synchronize()
            self._uploads = []

    def _uploaded(self):
        if not torch.cuda.is_current_stream_capturing():
            ev = torch.cuda.Event()
            ev.record()
            self._uploads.append(ev)

    def add_tn(self, U, V, C, alpha=1.0):
        """C[P,Q] += alpha * U^T V;  always taken (returns True): grouped when a table accepts it, else held back as a
        direct launch that `flush` / `run_bucket` issues."""
        self._host_tables_free()
        # the transpose-read kernel takes every problem (128x32 tiles for a rank <= 32 side); the register-transposing
        # kernel is the fallback (AQL_TN_OLD=1 prefers it, for comparison)
        kinds = ("n", "w") if TN_OLD else ("x", "w", "n")
        for k in kinds:
            nbytes, _, fill, _ = self.KINDS[k]
            slot = self.host[k].data_ptr() + self.n[k] * nbytes
            nblk = L.call_raw(fill, L.c_p(slot), L.ptr(U), U.stride(0), L.ptr(V), V.stride(0), U.shape[0], U.shape[1],
                              V.shape[1], float(alpha), L.ptr(C), C.stride(0), self.blk[k])
            if nblk > 0:
                self.items.append((C, k, nblk, None))
                self.n[k] += 1
                self.blk[k] += nblk
                self.keep += [U, V]
                return True
        self.items.append((C, "d", 0, (U, V, float(alpha))))
        self.keep += [U, V]
        return True

    def add_ds(self, dTs, T, dS, nb, rps, r):
        self._host_tables_free()
        slot = self.ds_host.data_ptr() + self.n_ds * self.DS_BYTES
        if not T.is_contiguous() or dTs.stride(1) != 1:
            return False
        nblk = L.call_raw("aql_ds_desc_fill_ld", L.c_p(slot), L.ptr(dTs), dTs.stride(0), L.ptr(T), nb, rps, r, L.ptr(dS), self.blk_ds)
        if nblk <= 0:
            return False
        self.n_ds += 1
        self.blk_ds += nblk
        self.keep += [dTs, T]
        return True

    def flush_ds(self):
        if self.n_ds:
            nbytes = self.n_ds * self.DS_BYTES
            self.ds_dev[:nbytes].copy_(self.ds_host[:nbytes], non_blocking=True)
            self._uploaded()
            L.call("aql_lora_ds_grouped", L.ptr(self.ds_dev), self.n_ds, self.blk_ds, L.stream_ptr())
        self.n_ds = self.blk_ds = 0

    def _launch(self, k, first, n, base, nblk):
        nbytes, _, _, entry = self.KINDS[k]
        lo, hi = first * nbytes, (first + n) * nbytes
        self.dev[k][lo:hi].copy_(self.host[k][lo:hi], non_blocking=True)
        self._uploaded()
        L.call(entry, L.ptr(self.dev[k]), first, n, base, nblk, L.stream_ptr())

    def flush_tn(self):
        for k in self.KINDS:
            if self.n[k]:
                self._launch(k, 0, self.n[k], 0, self.blk[k])
        for C, k, _, direct in self.items:
            if k == "d":
                gemm_tn_acc(direct[0], direct[1], C, direct[2])
        self.n = {k: 0 for k in self.KINDS}
        self.blk = {k: 0 for k in self.KINDS}
        self.items = []

    def flush(self):
        """One grouped launch per table for everything queued; then the post-phase of the weight-side LoRA sites (they read the
        per-sample dY^T X products of the first launch and queue their dA problems) and a second launch over the NEW descriptors
        only.  The descriptor tables are append-only within a step: under graph capture the H2D copy nodes re-read the pinned host
        tables at every replay, so a slot must never be rewritten between the two launches."""
        mark = {k: (0, 0) for k in self.KINDS}
        mark["items"] = 0

        def launch_new():
            for k in self.KINDS:
                first, base = mark[k]
                if self.n[k] > first:
                    self._launch(k, first, self.n[k] - first, base, self.blk[k] - base)
                mark[k] = (self.n[k], self.blk[k])
            for C, k, _, direct in self.items[mark["items"]:]:
                if k == "d":
                    gemm_tn_acc(direct[0], direct[1], C, direct[2])
            mark["items"] = len(self.items)

        launch_new()
        if self.post:
            post, self.post = self.post, []
            for fn in post:
                fn()
            launch_new()
        self.flush_ds()
        self.reset()

    # ---- bucketed form -------------------------------------------------------------------------------------
    def plan(self, flat_grad, nbuckets):
        """Sort the held-back problems by the offset of their output inside ``flat_grad`` and cut them into buckets.
        Rewrites the host descriptor tables in that order.  Returns [(lo, hi)] element ranges of flat_grad."""
        import numpy as np
        base, esz = flat_grad.data_ptr(), flat_grad.element_size()
        offs, sizes = [], []
        for C, _, _, _ in self.items:
            o = (C.data_ptr() - base) // esz
            if not (0 <= o and o + C.numel() <= flat_grad.numel() and C.is_contiguous()):
                raise L.AqlError("DeferredDW.plan: a weight gradient does not live in the flat gradient buffer")
            offs.append(o)
            sizes.append(C.numel())
        groups = plan_buckets(offs, sizes, nbuckets)
        tbl, old, slot_of, pos, blk = {}, {}, {}, {k: 0 for k in self.KINDS}, {k: 0 for k in self.KINDS}
        for k, (nbytes, _, _, _) in self.KINDS.items():
            tbl[k] = self.host[k].numpy()[:self.n[k] * nbytes].reshape(self.n[k], nbytes)
            old[k] = tbl[k].copy()
            idx = [i for i, it in enumerate(self.items) if it[1] == k]   # arrival order == table order
            slot_of[k] = {i: s for s, i in enumerate(idx)}
        self.buckets = []
        for lo, hi, idxs in groups:
            first, base_blk, direct = dict(pos), dict(blk), []
            for i in idxs:
                C, k, nblk, d = self.items[i]
                if k == "d":
                    direct.append((d[0], d[1], C, d[2]))
                    continue
                fb = self.KINDS[k][1]
                tbl[k][pos[k]] = old[k][slot_of[k][i]]
                tbl[k][pos[k], fb:fb + 4].view(np.int32)[0] = blk[k]     # first_block
                pos[k] += 1
                blk[k] += nblk
            self.buckets.append(dict(lo=lo, hi=hi, direct=direct,
                                     ranges={k: (first[k], pos[k] - first[k], base_blk[k], blk[k] - base_blk[k])
                                             for k in self.KINDS}))
        assert pos == self.n and blk == self.blk
        return [(b["lo"], b["hi"]) for b in self.buckets]

    def run_bucket(self, k):
        b = self.buckets[k]
        for kind, (first, n, base, nblk) in b["ranges"].items():
            if n:
                self._launch(kind, first, n, base, nblk)
        for U, V, C, alpha in b["direct"]:
            gemm_tn_acc(U, V, C, alpha)


class SplitDeferred:
    """Router in front of several DeferredDW tables for the overlapped data-parallel exchange: a weight gradient whose output
    lies inside [bounds[i], bounds[i+1]) -- region i of the flat gradient buffer, complete when backward finishes leg i
    (lora.LoraBank.cuts) -- is queued on ``tables[i]``; everything behind the last bound (and every dS reduction) on ``late``.
    The trainer flushes and all-reduces table i from the U-Net's backward hook i and ``late`` at the end of backward."""

    def __init__(self, tables, late, bounds):
        self.tables, self.late, self.bounds = list(tables), late, [int(b) for b in bounds]
        assert len(self.bounds) == len(self.tables) + 1

    def add_tn(self, U, V, C, alpha=1.0):
        p = C.data_ptr()
        tgt = self.late
        for i, t in enumerate(self.tables):
            if self.bounds[i] <= p < self.bounds[i + 1]:
                tgt = t
                break
        return tgt.add_tn(U, V, C, alpha)

    def add_ds(self, dTs, T, dS, nb, rps, r):
        return self.late.add_ds(dTs, T, dS, nb, rps, r)


REF_ROUNDING = os.environ.get("AQL_REF_ROUNDING", "0") == "1"   # see LoraLinearFn.forward
TN_OLD = False   # True = the older register-transposing weight-gradient kernel (tools/probe_tntr.py times it against the transpose-read one)
DEFERRED = None  # set by a trainer around backward (ppft.PPFTTrainer); None => every site launches its own kernels


# ------------------------------------------------------------------------------------ fused LoRA linear
def _lora_gemm_fused(x2d, w, a16, S16, rps, b16, bias, residual, T, Ts, y=None, row0=0):
    """One-launch rank-32 LoRA linear (aql_lora_gemm_fused).  Returns Y, or None when the shape belongs on the two-launch
    path (rank != 32, split-K shapes, narrow outputs; AQL_LORA_FUSED=0 disables it for comparison)."""
    if a16.shape[0] != 32 or os.environ.get("AQL_LORA_FUSED", "1") == "0":
        return None
    M, K = x2d.shape
    N = w.shape[0]
    if y is None:
        y = torch.empty(M, N, dtype=torch.bfloat16, device=x2d.device)
    rc = L.call_raw("aql_lora_gemm_fused", L.ptr(x2d), x2d.stride(0), L.ptr(w), w.stride(0), M, N, K, L.ptr(a16), L.ptr(S16),
                    rps, L.ptr(b16), L.ptr(bias), L.ptr(residual), 0 if residual is None else residual.stride(0), L.ptr(y),
                    y.stride(0), L.ptr(T), L.ptr(Ts), int(row0), L.stream_ptr())
    if rc == 100:
        return None
    L.check(rc, "aql_lora_gemm_fused")
    return y


def chain_fwd(x, ldx, M, rps, row0, S16, stages, rank=32):
    """aql_lora_chain_fwd (rank 32; rank 320: aql_lora_chain_fwd_r320): a row-resident chain of 320 -> 320 LoRA linears
    (csrc/aql_chain.hip).  ``stages``: list of dicts with the keys  W ldw bias Ad Bup T Ts res ldr out ldo keep ln gamma beta eps
    stats nout ldn nout_row0 oscale  (tensors or None; missing = None / 0; oscale: missing = 1)."""
    import ctypes
    n = len(stages)
    vp, lp_, ip, fp = ctypes.c_void_p * n, ctypes.c_long * n, ctypes.c_int * n, ctypes.c_float * n

    def ptrs(key):
        return vp(*[(None if st.get(key) is None else st[key].data_ptr()) for st in stages])

    def longs(key):
        return lp_(*[int(st.get(key) or 0) for st in stages])

    L.call("aql_lora_chain_fwd_r320" if rank == 320 else "aql_lora_chain_fwd", L.ptr(x), int(ldx), int(M), int(rps), int(row0), L.ptr(S16), n,
           ptrs("W"), longs("ldw"), ptrs("bias"), ptrs("Ad"), ptrs("Bup"), ptrs("T"), ptrs("Ts"), ptrs("res"), longs("ldr"),
           ptrs("out"), longs("ldo"), ip(*[int(st.get("keep") or 0) for st in stages]), ip(*[int(st.get("ln") or 0) for st in stages]),
           ptrs("gamma"), ptrs("beta"), fp(*[float(st.get("eps") or 0.0) for st in stages]), ptrs("stats"), ptrs("nout"),
           longs("ldn"), longs("nout_row0"), fp(*[float(st.get("oscale") or 1.0) for st in stages]), L.stream_ptr())


def chain_bwd(dy, lddy, M, rps, S16, stages, lns):
    """aql_lora_chain_bwd (csrc/aql_chain.hip): backward-data of a chain.  ``stages``: dicts  Wt ldw BupT AT dTs dT dX lddx keep;
    ``lns``: nstage + 1 entries (None or dict  x ldx stats gamma dres lddres out ldo): entry 0 = LayerNorm backward on the incoming
    gradient, entry g + 1 = behind stage g."""
    import ctypes
    n = len(stages)
    vp, lp_, ip = ctypes.c_void_p * n, ctypes.c_long * n, ctypes.c_int * n
    vq, lq = ctypes.c_void_p * (n + 1), ctypes.c_long * (n + 1)

    def ptrs(key):
        return vp(*[(None if st.get(key) is None else st[key].data_ptr()) for st in stages])

    def lptr(key):
        return vq(*[(None if (e is None or e.get(key) is None) else e[key].data_ptr()) for e in lns])

    def llong(key):
        return lq(*[int((e or {}).get(key) or 0) for e in lns])

    L.call("aql_lora_chain_bwd", L.ptr(dy), int(lddy), int(M), int(rps), L.ptr(S16), n,
           ptrs("Wt"), lp_(*[int(st.get("ldw") or 0) for st in stages]), ptrs("BupT"), ptrs("AT"), ptrs("dTs"), ptrs("dT"), ptrs("dX"),
           lp_(*[int(st.get("lddx") or 0) for st in stages]), ip(*[int(st.get("keep") or 0) for st in stages]),
           lptr("x"), llong("ldx"), lptr("stats"), lptr("gamma"), lptr("dres"), llong("lddres"), lptr("out"), llong("ldo"), L.stream_ptr())


_DOWN_COUNTERS = {}


def _down_counters(device):
    """Ticket counters of aql_lora_down_splitk (one int per 16-row block; zero between launches, the kernel resets them)."""
    k = _skey(device)
    c = _DOWN_COUNTERS.get(k)
    if c is None:
        c = _DOWN_COUNTERS[k] = torch.zeros(4096, dtype=torch.int32, device=device)
    return c


def lora_down(x, ldx, M, K, a16, rank, S16, rps, T, Ts):
    """T = X.A^T, Ts = T * S[row // rps] (aql_lora_down); few rows under a deep K split the K range over workgroups
    (aql_lora_down_splitk: partial sums in the shared workspace, last-arrival reduction)."""
    if rank == 32 and K >= 2048 and M <= 4096 and M <= 16 * 4096:
        ws, cnt = workspace(x.device), _down_counters(x.device)
        L.call("aql_lora_down_splitk", L.ptr(x), ldx, M, K, L.ptr(a16), rank, L.ptr(S16), rps, L.ptr(T), L.ptr(Ts), L.ptr(ws),
               ws.numel() * 4, L.ptr(cnt), cnt.numel() * 4, L.stream_ptr())
    else:
        L.call("aql_lora_down", L.ptr(x), ldx, M, K, L.ptr(a16), rank, L.ptr(S16), rps, L.ptr(T), L.ptr(Ts), None, None,
               L.stream_ptr())


def _lora_down_rows(xk, K, site, S16k, rps, Tk, Tsk, row0):
    """T = X.A^T, Ts = T*S for the rows that have a LoRA term (rows >= row0; row0 is a multiple of rps): the clean half of a twin
    batch is skipped -- its T / Ts rows are never read (aql_gemm_bf16_ex treats them as zeros)."""
    M = xk.shape[0] - row0
    lora_down(xk[row0:], xk.stride(0), M, K, site.a16, site.rank, S16k[row0 // rps:], rps, Tk[row0:], Tsk[row0:])


def _geglu_fused(x2d, packed, site, S16, rps, T, Ts, G, H, row0=0):
    """ff.net.0.proj (+ LoRA) + GEGLU in one launch into G [M,F] (and H [M,2F] unless None).  Returns "done", "down" (only
    the LoRA down product T / Ts was computed: finish with the plain GEMM + geglu kernel) or None (nothing done)."""
    M, K = x2d.shape
    F = packed.N // 2
    if F % 80 != 0 or os.environ.get("AQL_GEGLU_FUSED", "1") == "0":
        return None
    ldh = 2 * F
    if site is not None:
        rc = 100
        if site.rank == 32 and os.environ.get("AQL_LORA_FUSED", "1") != "0":
            rc = L.call_raw("aql_lora_gemm_fused_geglu", L.ptr(x2d), x2d.stride(0), L.ptr(packed.w), packed.w.stride(0), M, F, K,
                            L.ptr(site.a16), L.ptr(S16), rps, L.ptr(site.b16), L.ptr(packed.bias), L.ptr(H), ldh, L.ptr(G), F,
                            L.ptr(T), L.ptr(Ts), int(row0), L.stream_ptr())
        if rc == 100:   # two-launch form: skinny T product, then the GEGLU GEMM with Ts.Bup^T as a second K segment
            _lora_down_rows(x2d, K, site, S16, rps, T, Ts, row0)
            rc = L.call_raw("aql_gemm_bf16_geglu", L.ptr(x2d), x2d.stride(0), L.ptr(packed.w), packed.w.stride(0), M, F, K,
                            L.ptr(Ts), Ts.stride(0), L.ptr(site.b16), site.b16.stride(0), site.rank, L.ptr(packed.bias),
                            L.ptr(H), ldh, L.ptr(G), F, int(row0), L.stream_ptr())
            if rc == 100:
                return "down"
    else:
        rc = L.call_raw("aql_gemm_bf16_geglu", L.ptr(x2d), x2d.stride(0), L.ptr(packed.w), packed.w.stride(0), M, F, K, None, 0,
                        None, 0, 0, L.ptr(packed.bias), L.ptr(H), ldh, L.ptr(G), F, int(row0), L.stream_ptr())
        if rc == 100:
            return None
    L.check(rc, "geglu-fused linear")
    return "done"


# ----------------------------------------------------------------------- weight-side form of the LoRA linear
# OFF by default (AQL_WSIDE=1 enables it): measured round 4 on MI355X, BASELINE config 3 (rank 320, batch 8): 51.9 ms per step
# against 49.4 ms for the activation-side branch (profiles/r04_weight_side_lora.txt).  The FLOPs it removes (5.0 -> 1.1 GFLOP per
# square site and sample) are the EFFICIENT ones -- 32768-row GEMMs at 20-30 us each -- and what it adds are eight per-sample token
# reductions plus ~8 small launches per site.  Parity with the activation-side branch: 2-5e-3 relative L2 on every output and gradient.
_WSIDE = os.environ.get("AQL_WSIDE", "0") == "1"


def gemm_sw(A, B, N, K, out, Bs=None, sstride=0, srows=0, srow0=0, bias=None, residual=None, res_mod=0, G=None, geglu_F=0,
            c_row0=0, gb_h=None):
    """aql_gemm_bf16_sw: out[m] = A[m] . Wsel(m)^T (+ bias) (+ residual[m % res_mod]) with per-sample weights behind row srow0
    (see include/aqualora_hip.h).  Returns the raw status (100 = the GEGLU form has no tile for this shape)."""
    M = A.shape[0]
    ws = workspace(A.device)
    return L.call_raw("aql_gemm_bf16_sw", L.ptr(A), A.stride(0), L.ptr(B), B.stride(0), M, N, K, L.ptr(Bs), int(sstride), int(srows),
                      int(srow0), L.ptr(bias), L.ptr(residual), 0 if residual is None else residual.stride(0), int(res_mod),
                      L.ptr(out), 0 if out is None else out.stride(0), L.ptr(G), 0 if G is None else G.stride(0), int(geglu_F),
                      int(c_row0), L.ptr(gb_h), 0 if gb_h is None else gb_h.stride(0), L.ptr(ws), ws.numel() * 4, L.stream_ptr())


def wside_ok(packed, site, S16, rps, geglu=False):
    """Does the weight-side form (per-sample effective weights We_b = W + Bup.diag(S_b).A) pay for this site?
    Activation side: 6 rps r (N + K) FLOPs per sample over forward, backward-data and the two weight gradients.  Weight side:
    8 N K r (We, We^T, and the two r-wide products that turn dWe into dA / dBup / dS) + 2 rps N K (dWe = dY^T X).  Taken when the
    ratio exceeds 1.5 (rank 320 on the 320-channel level: 4.6x for the square projections, 2.6x for the feed-forward pair; the
    640-channel level sits at 1.3x and below and stays on the activation side), at ranks above 32 (rank 32 has the one-launch
    kernel), with whole 256-row tiles per sample, and only where the weight gradients are flushed in one piece (single GPU):
    the bucketed / overlapped data-parallel exchanges plan their buckets over direct weight-gradient outputs."""
    if not _WSIDE or site is None or S16 is None or site.rank <= 32 or rps % 256 != 0 or REF_ROUNDING:
        return False
    if geglu and (packed.N // 2) % 80 != 0:
        return False
    dfr = DEFERRED
    if dfr is not None and not getattr(dfr, "allow_post", False):
        return False
    r, N, K = site.rank, packed.N, packed.K
    if r % 8 or N % 8 or K % 8:
        return False
    return 6.0 * rps * r * (N + K) > 1.5 * (8.0 * N * K * r + 2.0 * rps * N * K)


def wside_weights(packed, site, S16):
    """(We [(b, n)][K], WeT [(b, k)][N]) bf16 for the B scale rows S16 [B, r]: two GEMMs over the rank, the frozen weight as the
    residual under every sample (res_mod).  The scaled stacks Bup * S_b / A^T * S_b are broadcast multiplies (plumbing)."""
    B, r = S16.shape
    N, K = packed.N, packed.K
    dev = S16.device
    bs = (site.b16.unsqueeze(0) * S16.unsqueeze(1)).reshape(B * N, r)        # [(b, n)][r]: kept for backward (the dA problem)
    We = torch.empty(B * N, K, dtype=torch.bfloat16, device=dev)
    L.check(gemm_sw(bs, site.at16, K, r, We, residual=packed.w, res_mod=N), "aql_gemm_bf16_sw (We)")
    ats = (site.at16.unsqueeze(0) * S16.unsqueeze(1)).reshape(B * K, r)      # [(b, k)][r]
    WeT = torch.empty(B * K, N, dtype=torch.bfloat16, device=dev)
    L.check(gemm_sw(ats, site.b16, N, r, WeT, residual=packed.wt, res_mod=K), "aql_gemm_bf16_sw (We^T)")
    return We, WeT, bs


def wside_backward(dy, x2d, WeT, bs, S16, packed, site, rps, ds_accum, want_dx, ret_ds, s_dtype, dx_prev, geglu_h=None):
    """Backward of the weight-side form: dX = dY . We_b (plain GEMM with per-sample weights; for ff.net.2 with the GEGLU-backward
    epilogue), dWe_b = dY_b^T X_b per sample (token-reduction problems on the trainer's grouped launch), then -- after that launch --
        P[(b, n)][j] = sum_k dWe_b[n][k] A[j][k]                      one GEMM over K
        dBup[n][j] += sum_b P[(b,n)][j] S_b[j]      dS_b[j] += sum_n Bup[n][j] P[(b,n)][j]        aql_wside_reduce
        dA[j][k]   += sum_(b,n) (Bup[n][j] S_b[j]) dWe_b[n][k]        one token-reduction problem over (b, n); `bs` from forward
    Returns (dX or None, dS in s_dtype or None)."""
    M, N, K, r = dy.shape[0], packed.N, packed.K, site.rank
    B = S16.shape[0]
    dev = dy.device
    dx = None
    if want_dx:
        if geglu_h is not None:       # ff.net.2: X = GEGLU(H); the epilogue turns d(activated) [M, K] into d(H) [M, 2K]
            dh = torch.empty(M, 2 * K, dtype=torch.bfloat16, device=dev)
            rc = gemm_sw(dy, packed.wt, K, N, dh, Bs=WeT, sstride=K * N, srows=rps, srow0=0, gb_h=geglu_h)
            if rc != 100:
                L.check(rc, "aql_gemm_bf16_sw (GEGLU backward)")
                dx = dh
        if dx is None:
            dx = torch.empty(M, K, dtype=torch.bfloat16, device=dev)
            L.check(gemm_sw(dy, packed.wt, K, N, dx, Bs=WeT, sstride=K * N, srows=rps, srow0=0, residual=dx_prev), "aql_gemm_bf16_sw (dX)")
            if geglu_h is not None:
                dh = torch.empty_like(geglu_h)
                L.call("aql_geglu_bwd", L.ptr(geglu_h), L.ptr(dx), M, geglu_h.shape[1] // 2, L.ptr(dh), L.stream_ptr())
                dx = dh
    acc = ds_accum
    want_ds = ret_ds or acc is not None
    dS = torch.zeros(B, r, dtype=torch.float32, device=dev) if (acc is None) else None
    ds_target = acc if acc is not None else dS
    dfr = DEFERRED
    immediate = dfr is None or not getattr(dfr, "allow_post", False) or (want_ds and acc is None)
    Gw = torch.zeros(B, N, K, dtype=torch.float32, device=dev)               # dWe_b = dY_b^T X_b, rows (b, n)
    for b in range(B):
        u, v, c = dy[b * rps:(b + 1) * rps], x2d[b * rps:(b + 1) * rps], Gw[b]
        if immediate or not dfr.add_tn(u, v, c):
            gemm_tn_acc(u, v, c)
    ga, gb = site.ga, site.gb

    def post():
        G16 = Gw.view(B * N, K).to(torch.bfloat16)
        P = gemm_bf16(G16, site.a16)                                         # [(b, n)][r]
        L.call("aql_wside_reduce", L.ptr(P), L.ptr(S16), L.ptr(site.b16), B, N, r, L.ptr(gb), gb.stride(0), L.ptr(ds_target),
               L.stream_ptr())
        if immediate or not dfr.add_tn(bs, G16, ga):
            gemm_tn_acc(bs, G16, ga)

    if immediate:
        post()
    else:
        dfr.post.append(post)
    if dS is not None:
        dS = dS.to(s_dtype) if ret_ds else None
    return dx, dS


class LoraLinearFn(torch.autograd.Function):
    """Y = X.W^T + b [+ ((X.A^T) * S[sample]).Bup^T] [+ residual]   on token-major X [M,K].

    Replaces CustomLoRACompatibleLinearforward + CustomLoRALinearLayerforward (reference
    utils/lora_modules.py:56-62, 9-26) and, for 1x1 convolutions on channels-last maps, the conv pair
    (:46-54, 28-44).  ``site`` is a lora.LoraSite (bf16 A, A^T, Bup, Bup^T and fp32 grad views) or None.
    ``geglu``: the host is ff.net.0.proj -- return  Y[:, :F] * gelu(Y[:, F:])  (GEGLU.forward, original_unet.py:727-729),
    applied in the GEMM epilogue; Y itself is kept only when a backward pass will need it (``want_h``).
    Twin batches (see `_Dual`): the kernels run on both halves, autograd sees and saves the second-half views.
    """

    @staticmethod
    def forward(ctx, x2d, packed, site, S, S16, rps, residual, geglu=False, want_h=True):
        _req(x2d, "lora_linear")
        M = x2d.shape[0]
        dev = x2d.device
        xk = _full(x2d)
        twin = xk is not None
        if not twin:
            xk = x2d
        resk = residual if (residual is None or not twin) else _need_full(residual, "the residual")
        ctx.packed, ctx.site, ctx.rps = packed, site, rps
        ctx.has_res = residual is not None
        use_lora = site is not None and S16 is not None
        ctx.use_lora = use_lora
        ctx.s_dtype = S.dtype if S is not None else None
        ctx.geglu = geglu
        T = Ts = Tk = Tsk = S16k = None
        ctx.wside = use_lora and wside_ok(packed, site, S16, rps, geglu)
        if ctx.wside:
            # weight-side form (rank 320 on the 320-channel level): per-sample effective weights, plain GEMMs, no T / Ts
            ctx.ds_accum = getattr(S, "_aql_ds_accum", None)
            We, WeT, bs = wside_weights(packed, site, S16)
            row0 = M if twin else 0     # twin batch: rows [0, M) are the clean pass and use the frozen W
            F = packed.N // 2
            Mk = xk.shape[0]
            if geglu:
                assert residual is None
                yk, y = _alloc((M, F), torch.bfloat16, dev, twin)
                hk, h = _alloc((M, packed.N), torch.bfloat16, dev, twin) if want_h else (None, None)
                rc = gemm_sw(xk, packed.w, packed.N, packed.K, hk, Bs=We, sstride=packed.N * packed.K, srows=rps, srow0=row0,
                             bias=packed.bias, G=yk, geglu_F=F, c_row0=row0 if _TWIN_SKIP else 0)
                if rc == 100:           # no 160-wide tile: plain GEMM into H, then the activation kernel
                    if hk is None:
                        hk, h = _alloc((M, packed.N), torch.bfloat16, dev, twin)
                    L.check(gemm_sw(xk, packed.w, packed.N, packed.K, hk, Bs=We, sstride=packed.N * packed.K, srows=rps,
                                    srow0=row0, bias=packed.bias), "aql_gemm_bf16_sw")
                    L.call("aql_geglu_fwd", L.ptr(hk), Mk, F, L.ptr(yk), L.stream_ptr())
                else:
                    L.check(rc, "aql_gemm_bf16_sw (GEGLU)")
                ctx.save_for_backward(x2d, WeT, bs, S16, h)
                return y
            yk, y = _alloc((M, packed.N), torch.bfloat16, dev, twin)
            L.check(gemm_sw(xk, packed.w, packed.N, packed.K, yk, Bs=We, sstride=packed.N * packed.K, srows=rps, srow0=row0,
                            bias=packed.bias, residual=resk), "aql_gemm_bf16_sw")
            ctx.save_for_backward(x2d, WeT, bs, S16, None)
            return y
        if use_lora:
            r = site.rank
            S16k = _need_full(S16, "the LoRA scale") if twin else S16
            Tk, T = _alloc((M, r), torch.bfloat16, dev, twin)
            Tsk, Ts = _alloc((M, r), torch.bfloat16, dev, twin)
            # trainers can hand in one persistent fp32 accumulator for dS (shared by all 192 sites)
            ctx.ds_accum = getattr(S, "_aql_ds_accum", None)
        F = packed.N // 2
        if use_lora and REF_ROUNDING:
            # Debug mode (AQL_REF_ROUNDING=1): the reference's rounding points under bf16 autocast (utils/lora_modules.py:13-19,
            # 56-62) instead of the single fp32 accumulator -- T = bf16(x.A^T); Ts = bf16(T*S); lora = bf16(Ts.Bup^T);
            # y = bf16(bf16(x.W^T + b) + lora); the host's residual is one more bf16 add.  Three launches; for bit-level
            # comparisons of one site against an autocast run of the reference, not for training.
            yk, y = _alloc((M, packed.N), torch.bfloat16, dev, twin)
            _lora_down_rows(xk, packed.K, site, S16k, rps, Tk, Tsk, 0)
            base = gemm_bf16(xk, packed.w, packed.bias)
            gemm_bf16(Tsk, site.b16, None, residual=base, out=yk)
            if resk is not None:
                yk.add_(resk)
            if geglu:
                gk, g_out = _alloc((M, F), torch.bfloat16, dev, twin)
                L.call("aql_geglu_fwd", L.ptr(yk), yk.shape[0], F, L.ptr(gk), L.stream_ptr())
                ctx.save_for_backward(x2d, T, Ts, S16, y)
                return g_out
            ctx.save_for_backward(x2d, T, Ts, S16, None)
            return y
        yk, y = _alloc((M, F if geglu else packed.N), torch.bfloat16, dev, twin)
        hk = h = None
        done = None
        row0 = M if (twin and _TWIN_SKIP) else 0   # twin batch: rows [0, M) are the clean pass (all-zero scale): no LoRA term, no backward
        if geglu:
            assert residual is None
            if want_h:
                hk, h = _alloc((M, packed.N), torch.bfloat16, dev, twin)
            done = _geglu_fused(xk, packed, site if use_lora else None, S16k, rps, Tk, Tsk, yk, hk, row0)
        if done != "done":
            if geglu and hk is None:
                hk, h = _alloc((M, packed.N), torch.bfloat16, dev, twin)
            out = hk if geglu else yk
            if use_lora:
                ok = None
                if done != "down":
                    ok = _lora_gemm_fused(xk, packed.w, site.a16, S16k, rps, site.b16, packed.bias, resk, Tk, Tsk, out, row0)
                if ok is None:   # two-launch form: skinny T product, then the GEMM with Ts.Bup^T as a second K segment
                    if done != "down":
                        _lora_down_rows(xk, packed.K, site, S16k, rps, Tk, Tsk, row0)
                    gemm_bf16(xk, packed.w, packed.bias, Tsk, site.b16, residual=resk, out=out, lora_row0=row0)
            else:
                gemm_bf16(xk, packed.w, packed.bias, residual=resk, out=out)
            if geglu:   # unfused tail (shapes without a 160-wide tile): separate activation kernel
                L.call("aql_geglu_fwd", L.ptr(hk), hk.shape[0], F, L.ptr(yk), L.stream_ptr())
        if use_lora:
            ctx.save_for_backward(x2d, T, Ts, S16, h if geglu else None)
        else:
            ctx.save_for_backward(x2d, h if geglu else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        packed, site = ctx.packed, ctx.site
        M = dy.shape[0]
        dS = None
        if ctx.geglu:   # d(pre-activation) from d(activated): dy becomes [M, 2F]
            h = ctx.saved_tensors[-1]
            dh = torch.empty_like(h)
            L.call("aql_geglu_bwd", L.ptr(h), L.ptr(dy), M, h.shape[1] // 2, L.ptr(dh), L.stream_ptr())
            dy = dh
        if ctx.use_lora:
            x2d, T, Ts, S16 = ctx.saved_tensors[:4]
            dx, dS = _lora_backward(dy, x2d, T, Ts, S16, packed, site, ctx.rps, ctx.ds_accum, ctx.needs_input_grad[0],
                                    ctx.needs_input_grad[3], ctx.s_dtype, None, wside=ctx.wside)
        else:
            dx = gemm_bf16(dy, packed.wt) if ctx.needs_input_grad[0] else None
        return dx, None, None, dS, None, None, (dy if ctx.has_res else None), None, None


def _lora_backward(dy, x2d, T, Ts, S16, packed, site, rps, ds_accum, want_dx, ret_ds, s_dtype, dx_prev, geglu_h=None, wside=False):
    """Backward of one LoRA linear  Y = X.W^T + ((X.A^T)*S).Bup^T :  dTs = dY.Bup, dT = dTs*S, dX = dY.W + dT.A (+ dx_prev, added
    in the GEMM epilogue), dS += rowsum(dTs*T) per sample, and the weight gradients dBup += dY^T.Ts, dA += dT^T.X (queued on the
    trainer's DeferredDW when there is one).  Returns (dX or None, dS in s_dtype or None).
    ``geglu_h`` (ff.net.2 only): X is GEGLU(H); the returned gradient is d(H) [M, 2F] -- the GEGLU backward runs in the epilogue
    of the same launch (aql_lora_gemm_fused_geglu_bwd), or as aql_geglu_bwd behind the unfused forms."""
    if wside:   # weight-side form: T holds We^T [(b, k)][N], Ts the scaled up-matrices [(b, n)][r] (LoraLinearFn.forward)
        return wside_backward(dy, x2d, T, Ts, S16, packed, site, rps, ds_accum, want_dx, ret_ds, s_dtype, dx_prev, geglu_h)
    M = dy.shape[0]
    r = site.rank
    dS = None
    dTs = torch.empty(M, r, dtype=torch.bfloat16, device=dy.device)
    dT = torch.empty_like(dTs)
    nb = S16.shape[0]
    acc = ds_accum
    want_ds = ret_ds or acc is not None
    if want_ds and acc is None:
        dS = torch.zeros(nb, r, dtype=torch.float32, device=dy.device)
    dfr = DEFERRED
    ds_target = acc if acc is not None else dS
    ds_deferred = want_ds and dfr is not None and acc is not None
    dx = None
    fused_gb = False
    if want_dx and geglu_h is not None and r == 32 and os.environ.get("AQL_LORA_FUSED", "1") != "0" \
            and os.environ.get("AQL_GEGLU_BWD_FUSED", "1") != "0":
        F = packed.K
        dh = torch.empty(M, 2 * F, dtype=torch.bfloat16, device=dy.device)
        rc = L.call_raw("aql_lora_gemm_fused_geglu_bwd", L.ptr(dy), dy.stride(0), L.ptr(packed.wt), packed.wt.stride(0), M, F,
                        packed.N, L.ptr(site.bt16), L.ptr(S16), rps, L.ptr(site.at16), L.ptr(geglu_h), geglu_h.stride(0),
                        L.ptr(dh), 2 * F, L.ptr(dTs), L.ptr(dT), L.stream_ptr())
        if rc != 100:
            L.check(rc, "aql_lora_gemm_fused_geglu_bwd")
            dx, fused_gb = dh, True
    if want_dx and dx is None:   # dTs = dY.Bup, dT = dTs * S, dX = dY.W + dT.A in one launch
        dx = _lora_gemm_fused(dy, packed.wt, site.bt16, S16, rps, site.at16, None, dx_prev, dTs, dT)
    if dx is not None:
        if want_ds and not ds_deferred:
            L.call("aql_lora_ds", L.ptr(dTs), L.ptr(T), nb, rps, r, L.ptr(ds_target), L.stream_ptr())
    else:
        if want_ds and not ds_deferred:
            L.call("aql_lora_down", L.ptr(dy), dy.stride(0), M, packed.N, L.ptr(site.bt16), r, L.ptr(S16), rps,
                   L.ptr(dTs), L.ptr(dT), L.ptr(T), L.ptr(ds_target), L.stream_ptr())
        else:
            lora_down(dy, dy.stride(0), M, packed.N, site.bt16, r, S16, rps, dTs, dT)
        if want_dx and geglu_h is not None and dx_prev is None and os.environ.get("AQL_GEGLU_BWD_FUSED", "1") != "0":
            # ff.net.2 at rank != 32: the GEGLU backward in the epilogue of the two-K-segment GEMM
            F = packed.K
            dh = torch.empty(M, 2 * F, dtype=torch.bfloat16, device=dy.device)
            ws = workspace(dy.device)
            rc = L.call_raw("aql_gemm_bf16_geglu_bwd", L.ptr(dy), dy.stride(0), L.ptr(packed.wt), packed.wt.stride(0), M, F,
                            packed.N, L.ptr(dT), dT.stride(0), L.ptr(site.at16), site.at16.stride(0), r, L.ptr(geglu_h),
                            geglu_h.stride(0), L.ptr(dh), 2 * F, L.ptr(ws), ws.numel() * 4, L.stream_ptr())
            if rc != 100:
                L.check(rc, "aql_gemm_bf16_geglu_bwd")
                dx, fused_gb = dh, True
        if want_dx and dx is None:
            dx = gemm_bf16(dy, packed.wt, None, dT, site.at16, residual=dx_prev)
    if ds_deferred and not dfr.add_ds(dTs, T, ds_target, nb, rps, r):
        L.call("aql_lora_ds", L.ptr(dTs), L.ptr(T), nb, rps, r, L.ptr(ds_target), L.stream_ptr())
    # dBup[N,r] += dY^T Ts ; dA[r,K] += dT^T X   (grouped at the end of backward when a trainer defers them)
    if dfr is None or not dfr.add_tn(dy, Ts, site.gb):
        gemm_tn_acc(dy, Ts, site.gb)
    if dfr is None or not dfr.add_tn(dT, x2d, site.ga):
        gemm_tn_acc(dT, x2d, site.ga)
    if dS is not None:
        dS = dS.to(s_dtype) if ret_ds else None
    if geglu_h is not None and dx is not None and not fused_gb:   # unfused tail: d(activated) -> d(pre-activation)
        dh = torch.empty_like(geglu_h)
        L.call("aql_geglu_bwd", L.ptr(geglu_h), L.ptr(dx), M, geglu_h.shape[1] // 2, L.ptr(dh), L.stream_ptr())
        dx = dh
    return dx, dS


class _Saved:
    """stand-in for an autograd ctx when one Function composes the forward of another"""

    def save_for_backward(self, *t):
        self.saved = t


class FeedForwardFn(torch.autograd.Function):
    """diffusers FeedForward (GEGLU proj -> GEGLU -> net.2; scripts/lib/original_unet.py:727-760) with LoRA on both linears as
    ONE autograd node: forward = the ff.net.0 launch with the GEGLU epilogue, then the ff.net.2 launch (residual in its
    epilogue); backward = the ff.net.2 backward-data launch whose epilogue applies the GEGLU backward, then the ff.net.0
    backward-data launch -- no stand-alone GEGLU kernel in either direction, and d(activated) never touches HBM."""

    @staticmethod
    def forward(ctx, x2d, packed0, site0, packed2, site2, S, S16, rps, residual):
        c0, c2 = _Saved(), _Saved()
        g = LoraLinearFn.forward(c0, x2d, packed0, site0, S, S16, rps, None, True, True)
        y = LoraLinearFn.forward(c2, g, packed2, site2, S, S16, rps, residual, False, True)
        ctx.p0, ctx.s0, ctx.p2, ctx.s2, ctx.rps = packed0, site0, packed2, site2, rps
        ctx.ds_accum, ctx.s_dtype, ctx.has_res = c0.ds_accum, S.dtype, residual is not None
        ctx.w0, ctx.w2 = c0.wside, c2.wside
        _, T0, Ts0, _, h = c0.saved
        _, T2, Ts2, _, _ = c2.saved
        ctx.save_for_backward(x2d, T0, Ts0, S16, h, g, T2, Ts2)
        return y

    @staticmethod
    def backward(ctx, dy):
        x2d, T0, Ts0, S16, h, g, T2, Ts2 = ctx.saved_tensors
        dy = dy.contiguous()
        ret_ds = ctx.needs_input_grad[5]
        dh, dS2 = _lora_backward(dy, g, T2, Ts2, S16, ctx.p2, ctx.s2, ctx.rps, ctx.ds_accum, True, ret_ds, ctx.s_dtype, None,
                                 geglu_h=h, wside=ctx.w2)
        dx, dS0 = _lora_backward(dh, x2d, T0, Ts0, S16, ctx.p0, ctx.s0, ctx.rps, ctx.ds_accum, ctx.needs_input_grad[0], ret_ds,
                                 ctx.s_dtype, None, wside=ctx.w0)
        dS = None
        if dS0 is not None or dS2 is not None:
            dS = dS0 if dS2 is None else (dS2 if dS0 is None else dS0 + dS2)
        return dx, None, None, None, None, dS, None, None, (dy if ctx.has_res else None)


def feed_forward(x2d, packed0, site0, packed2, site2, S, S16, rps, residual=None):
    return FeedForwardFn.apply(x2d, packed0, site0, packed2, site2, S, S16, rps, residual)


# --------------------------------------------------------------------------- grouped LoRA linears (shared input)
def adjacent(tensors):
    """True when the tensors lie back to back in memory in this order (one contiguous stacked matrix)."""
    for a, b in zip(tensors, tensors[1:]):
        if not (a.is_contiguous() and b.is_contiguous()) or a.data_ptr() + a.numel() * a.element_size() != b.data_ptr():
            return False
    return True


_GROUPED = True    # False = every grouped rank-32 launch as per-site launches (module attribute; the environment hook left in round 6)
_KGROUPS = True   # False = three chained backward-data launches for q | k | v (module attribute; the environment hook left in round 6)


class GroupedLoraFn(torch.autograd.Function):
    """G rank-32 LoRA linears that read the SAME input, as one launch (aql_lora_gemm_fused_grouped): q|k|v of a self-attention
    (original_unet.py:688-704) or the k|v projections of the text states of all cross-attentions.  ``wcat`` [sum N_g, K] is the
    stacked frozen weight; the sites' bf16 A / Bup copies must be stacked in memory in group order (lora.LoraBank lays them out
    so); returns the G outputs as column views of one [M, sum N_g] buffer.  Backward: one fused backward-data launch per group,
    each adding the previous group's dX in its epilogue (no separate accumulation kernels), weight gradients queued as usual."""

    @staticmethod
    def forward(ctx, x2d, wcat, packs, sites, S, S16, rps):
        _req(x2d, "grouped lora_linear")
        M, K = x2d.shape
        G = len(sites)
        dev = x2d.device
        xk = _full(x2d)
        twin = xk is not None
        if not twin:
            xk = x2d
        S16k = _need_full(S16, "the LoRA scale") if twin else S16
        cols = [0]
        for p in packs:
            cols.append(cols[-1] + p.N)
        N = cols[-1]
        Mk = xk.shape[0]
        yk = torch.empty(Mk, N, dtype=torch.bfloat16, device=dev)
        Tk = torch.empty(G, Mk, 32, dtype=torch.bfloat16, device=dev)
        Tsk = torch.empty_like(Tk)
        if twin:
            DUAL.register(yk)
        import ctypes
        cs = (ctypes.c_int * (G + 1))(*cols)
        rc = L.call_raw("aql_lora_gemm_fused_grouped", L.ptr(xk), xk.stride(0), L.ptr(wcat), wcat.stride(0), Mk, N, K, G, cs,
                        L.ptr(sites[0].a16), L.ptr(S16k), rps, L.ptr(sites[0].b16), None, L.ptr(yk), N, L.ptr(Tk), L.ptr(Tsk),
                        M if (twin and _TWIN_SKIP) else 0, L.stream_ptr())
        L.check(rc, "aql_lora_gemm_fused_grouped")
        y = yk[M:] if twin else yk
        ctx.packs, ctx.sites, ctx.rps, ctx.s_dtype = packs, sites, rps, S.dtype
        ctx.ds_accum = getattr(S, "_aql_ds_accum", None)
        T = Tk[:, M:] if twin else Tk
        Ts = Tsk[:, M:] if twin else Tsk
        ctx.save_for_backward(x2d, T, Ts, S16)
        return tuple(y[:, cols[g]:cols[g + 1]] for g in range(G))

    @staticmethod
    def backward(ctx, *dys):
        x2d, T, Ts, S16 = ctx.saved_tensors
        dx, dS_sum = None, None
        if (not ctx.needs_input_grad[0] and ctx.ds_accum is not None and DEFERRED is not None
                and all(dy is not None for dy in dys) and _GROUPED):
            # input without gradient (the text states): only dTs / dT are needed (for dS, dA, dB) -- ONE grouped skinny launch
            # for all groups instead of one per group, everything else is bookkeeping for the deferred grouped launches
            import ctypes
            G = len(dys)
            M = x2d.shape[0]
            dys = [dy.contiguous() for dy in dys]
            dTs = torch.empty(G, M, 32, dtype=torch.bfloat16, device=x2d.device)
            dT = torch.empty_like(dTs)
            Xp = (ctypes.c_void_p * G)(*[dy.data_ptr() for dy in dys])
            Ap = (ctypes.c_void_p * G)(*[s_.bt16.data_ptr() for s_ in ctx.sites])
            Kp = (ctypes.c_int * G)(*[dy.shape[1] for dy in dys])
            L.call("aql_lora_down_grouped", G, Xp, Ap, Kp, M, L.ptr(S16), ctx.rps, L.ptr(dTs), L.ptr(dT), L.stream_ptr())
            nb = S16.shape[0]
            for g, dy in enumerate(dys):
                site = ctx.sites[g]
                if not DEFERRED.add_ds(dTs[g], T[g], ctx.ds_accum, nb, ctx.rps, 32):
                    L.call("aql_lora_ds", L.ptr(dTs[g]), L.ptr(T[g]), nb, ctx.rps, 32, L.ptr(ctx.ds_accum), L.stream_ptr())
                DEFERRED.add_tn(dy, Ts[g], site.gb)
                DEFERRED.add_tn(dT[g], x2d, site.ga)
            return None, None, None, None, None, None, None
        dx, dS = _grouped_backward(dys, x2d, T, Ts, S16, ctx.packs, ctx.sites, ctx.rps, ctx.ds_accum, ctx.needs_input_grad[0],
                                   ctx.needs_input_grad[4], ctx.s_dtype)
        return dx, None, None, None, dS, None, None


def _grouped_backward(dys, x2d, T, Ts, S16, packs, sites, rps, ds_accum, want_dx, ret_ds, s_dtype, rep=None):
    """Backward of G LoRA linears that share their input x2d (q | k | v): dX = sum_g dX_g, weight gradients queued.  T / Ts are indexable
    by group.  ``rep``: the G-fold repeated scale rows (rank > 32, ChainFn.forward made them) -- enables the two-launch grouped form.
    -> (dX or None, dS or None)."""
    dx, dS_sum = None, None
    G = len(dys)
    if (rep is not None and G == 3 and ds_accum is not None and DEFERRED is not None and not ret_ds
            and grouped_wide_ok(x2d, packs, sites, S16, need_dx=want_dx)):
        dcat = packed_columns(dys, x2d.shape[0], packs[0].N)
        if dcat is not None:   # [dQ | dK | dV] as the attention backward wrote it (pack_grads): two launches for the three sites
            return wide_grouped_backward(dcat, x2d, T, Ts, S16, rep, packs, sites, rps, ds_accum, want_dx), None
    if (want_dx and 2 <= G <= 3 and all(dy is not None for dy in dys) and ds_accum is not None
            and DEFERRED is not None and _KGROUPS and all(s_.rank == 32 for s_ in sites)):   # (dS goes to the trainer's accumulator)
        # q | k | v backward-data as ONE launch: dX = sum_g (dY_g.W_g + ((dY_g.Bup_g) * S).A_g), accumulators in registers
        import ctypes
        M, Kin = x2d.shape
        dys = [dy.contiguous() for dy in dys]
        dTs = torch.empty(G, M, 32, dtype=torch.bfloat16, device=x2d.device)
        dT = torch.empty_like(dTs)
        dx = torch.empty(M, Kin, dtype=torch.bfloat16, device=x2d.device)
        vp, lp_, ip = ctypes.c_void_p * G, ctypes.c_long * G, ctypes.c_int * G
        rc = L.call_raw("aql_lora_gemm_fused_kgroups", G, vp(*[dy.data_ptr() for dy in dys]), lp_(*[dy.stride(0) for dy in dys]),
                        vp(*[p.wt.data_ptr() for p in packs]), lp_(*[p.wt.stride(0) for p in packs]),
                        ip(*[dy.shape[1] for dy in dys]), vp(*[s_.bt16.data_ptr() for s_ in sites]),
                        vp(*[s_.at16.data_ptr() for s_ in sites]), M, Kin, L.ptr(S16), rps, None, 0, L.ptr(dx), Kin,
                        vp(*[dTs[g].data_ptr() for g in range(G)]), vp(*[dT[g].data_ptr() for g in range(G)]), L.stream_ptr())
        if rc != 100:
            L.check(rc, "aql_lora_gemm_fused_kgroups")
            nb = S16.shape[0]
            for g, dy in enumerate(dys):
                site = sites[g]
                if not DEFERRED.add_ds(dTs[g], T[g], ds_accum, nb, rps, 32):
                    L.call("aql_lora_ds", L.ptr(dTs[g]), L.ptr(T[g]), nb, rps, 32, L.ptr(ds_accum), L.stream_ptr())
                if not DEFERRED.add_tn(dy, Ts[g], site.gb):
                    gemm_tn_acc(dy, Ts[g], site.gb)
                if not DEFERRED.add_tn(dT[g], x2d, site.ga):
                    gemm_tn_acc(dT[g], x2d, site.ga)
            return dx, None
        dx = None
    for g, dy in enumerate(dys):
        if dy is None:
            continue
        dy = dy.contiguous()
        dx_g, dS = _lora_backward(dy, x2d, T[g], Ts[g], S16, packs[g], sites[g], rps, ds_accum, want_dx, ret_ds, s_dtype, dx)
        if dx_g is not None:
            dx = dx_g
        if dS is not None:
            dS_sum = dS if dS_sum is None else dS_sum + dS
    return dx, dS_sum


# ------------------------------------------------------------- q | k | v at a LoRA rank above 32 (BASELINE config 3: rank 320)
GROUPED_WIDE = os.environ.get("AQL_GROUPED_WIDE", "1") != "0"   # A/B hook: 0 = q | k | v (and the text k | v) as two-launch LoRA linears per site
# dS of the grouped sites: one [nb, G r] fp32 accumulator per group size G (one column block per member), kept ON the trainer's dS
# accumulator tensor (attribute `_aql_dsg`: {G: [tensor, dirty]}) so that it lives and dies with it; fold_ds3 adds the blocks up


def _rep_g(S, S16k, G):
    """[S16k | ... | S16k] ([nb, G r] bf16): the row scale of G stacked down products.  One small launch per step and G, cached on the
    scale tensor autograd sees (shared by the 16 blocks)."""
    c = getattr(S, "_aql_rep", None)
    if c is None:
        c = S._aql_rep = {}
    key = (G, S16k.untyped_storage().data_ptr(), S16k.shape[0])
    t = c.get(key)
    if t is None:
        t = c[key] = S16k.repeat(1, G).contiguous()
    return t


def fold_ds3(ds_accum):
    """dS += the column blocks of the grouped sites' accumulators (then zero them for the next step).  Called by the trainer after the
    deferred dS launch, before S.backward(ds_accum)."""
    nb, r = ds_accum.shape
    for G, t in getattr(ds_accum, "_aql_dsg", {}).items():
        if t[1]:
            ds_accum.add_(t[0].view(nb, G, r).sum(dim=1))
            t[0].zero_()
            t[1] = False


def grouped_wide_ok(x2d, packs, sites, S16, need_dx=True):
    """aql_gemm_bf16_grouped serves G = 2 or 3 hosts that read the same rows: one rank r > 32 (a multiple of 320: every tile width
    divides the column groups), equal hosts without bias whose width is a multiple of 320, bf16 copies laid out by lora.LoraBank
    (A and Bup stacked, Bup^T stacked; with an input gradient also A^T as column blocks of one [K, G r] matrix)."""
    G = len(sites)
    if not GROUPED_WIDE or S16 is None or REF_ROUNDING or G not in (2, 3) or any(s is None for s in sites):
        return False
    r, C, K = sites[0].rank, packs[0].N, packs[0].K
    if r <= 32 or r % 320 or C % 320 or K % 8 or any(s.rank != r for s in sites) or any(p.N != C or p.K != K or p.bias is not None for p in packs):
        return False
    if any(wside_ok(p, s_, S16, 256) for p, s_ in zip(packs, sites)):
        return False
    if not (adjacent([s.a16 for s in sites]) and adjacent([s.b16 for s in sites]) and adjacent([s.bt16 for s in sites])):
        return False
    if not need_dx:
        return True
    at = [s.at16 for s in sites]
    e = at[0].element_size()
    return all(t.stride(0) == G * r and t.stride(1) == 1 and t.data_ptr() == at[0].data_ptr() + g * r * e for g, t in enumerate(at))


def wcat_t(packs):
    """[W_1^T | .. | W_G^T] ([K, G C], column blocks) of frozen packed hosts: the weight-side operand of
    dX = [dY_1 | .. | dY_G].[W_1 | .. | W_G] in the grouped backward.  Built once, cached on the first host's packed copy."""
    c = getattr(packs[0], "_aql_wcat_t", None)
    if c is None or len(c[0]) != len(packs) or any(a is not b for a, b in zip(c[0], packs)):
        c = (tuple(packs), torch.cat([p.wt for p in packs], dim=1).contiguous())
        packs[0]._aql_wcat_t = c
    return c[1]


def packed_columns(dys, M, C):
    """dys as the column blocks of ONE [M, G C] buffer (what AttentionFn's pack_grads writes), viewed in place -- or None."""
    G, d0 = len(dys), dys[0]
    if d0 is not None and all(d is not None and d.shape == (M, C) and d.stride() == (G * C, 1) and
                              d.untyped_storage().data_ptr() == d0.untyped_storage().data_ptr() and
                              d.storage_offset() == d0.storage_offset() + g * C for g, d in enumerate(dys)):
        return d0.as_strided((M, G * C), (G * C, 1), d0.storage_offset())
    return None


def wide_grouped_backward(dcat, x2d, T, Ts, S16, rep, packs, sites, rps, ds_accum, want_dx):
    """Backward of G rank-r (r > 32) LoRA linears that read x2d, whose output gradients are the column blocks of dcat [M, G C] and whose
    saved T / Ts are per-site dense tensors (the q | k | v stages of a row-resident chain, ops.ChainFn): the two launches of
    GroupedWideFn.backward; dS and the weight gradients are queued per site (dTs / dT as strided column views).  -> dX or None."""
    G, r, C = len(sites), sites[0].rank, packs[0].N
    M, K = x2d.shape
    dev = x2d.device
    dTs = torch.empty(M, G * r, dtype=torch.bfloat16, device=dev)
    dT = torch.empty_like(dTs)
    L.call("aql_gemm_bf16_grouped", L.ptr(dcat), G * C, L.ptr(sites[0].bt16), C, M, G * r, C, None, 0, None, 0, 0, r, C, 0, None,
           None, 0, L.ptr(dTs), G * r, L.ptr(dT), G * r, L.ptr(rep), rps, 0, None, 0, L.stream_ptr())
    dx = None
    if want_dx:
        at0 = sites[0].at16
        acatT = at0.as_strided((K, G * r), (G * r, 1), at0.storage_offset())
        dx = gemm_bf16(dcat, wcat_t(packs), None, dT, acatT)
    nb = S16.shape[0]
    for g, site in enumerate(sites):
        dTs_g, dT_g = dTs[:, g * r:(g + 1) * r], dT[:, g * r:(g + 1) * r]
        if not DEFERRED.add_ds(dTs_g, T[g], ds_accum, nb, rps, r):
            L.call("aql_lora_ds", L.ptr(dTs_g.contiguous()), L.ptr(T[g]), nb, rps, r, L.ptr(ds_accum), L.stream_ptr())
        DEFERRED.add_tn(dcat[:, g * C:(g + 1) * C], Ts[g], site.gb)
        DEFERRED.add_tn(dT_g, x2d, site.ga)
    return dx


class GroupedWideFn(torch.autograd.Function):
    """G = 3: q | k | v of a self-attention, G = 2: k | v of a text-state attention, with a rank-r watermark LoRA, r > 32
    (utils/lora_modules.py:9-26, 56-62 on the hosts of scripts/lib/original_unet.py:688-704), as FOUR launches where the per-site path
    runs 4 G:
        forward    [T_1 | .. | T_G] = X.[A_1; ..; A_G]^T,  Ts = T * [S | .. | S]                 aql_lora_down, r' = G r
                   [y_1 | .. | y_G] = X.[W_1; ..; W_G]^T + Ts_g.Bup_g^T                          aql_gemm_bf16_grouped (grp_n = C, A2 offset r)
        backward   [dTs_1 | .. | dTs_G] = dY_g.Bup_g,  dT = dTs * [S | .. | S]                   aql_gemm_bf16_grouped (grp_n = r, A offset C)
                   dX = [dY_1 | ..].[W_1 | ..] + [dT_1 | ..].[A_1; ..]   (not for the text states)   aql_gemm_bf16_ex, K = G C and G r
    Forward outputs, T and Ts have the bits of the per-site launches (same accumulation order per element); dX is ONE fp32
    accumulation where the per-site path rounds on the way (more accurate, not bit-identical).  [dY_1 | .. | dY_G] is the buffer the
    attention backward wrote (AttentionFn, pack_grads), read in place.  dS goes to a [nb, G r] accumulator folded once per step
    (fold_ds3); the 2 G weight gradients are queued on the trainer's grouped launch as strided column views."""

    @staticmethod
    def forward(ctx, x2d, wcat, wcatT, packs, sites, S, S16, rps):
        _req(x2d, "grouped wide lora_linear")
        M, K = x2d.shape
        G, r, C = len(sites), sites[0].rank, packs[0].N
        dev = x2d.device
        xk = _full(x2d)
        twin = xk is not None
        if not twin:
            xk = x2d
        S16k = _need_full(S16, "the LoRA scale") if twin else S16
        Mk = xk.shape[0]
        row0 = M if (twin and _TWIN_SKIP) else 0
        rep = _rep_g(S, S16k, G)
        Tk = torch.empty(Mk, G * r, dtype=torch.bfloat16, device=dev)
        Tsk = torch.empty_like(Tk)
        lora_down(xk[row0:], xk.stride(0), Mk - row0, K, sites[0].a16, G * r, rep[row0 // rps:], rps, Tk[row0:], Tsk[row0:])
        yk = torch.empty(Mk, G * C, dtype=torch.bfloat16, device=dev)
        if twin:
            DUAL.register(yk)
        ws = workspace(dev)
        L.call("aql_gemm_bf16_grouped", L.ptr(xk), xk.stride(0), L.ptr(wcat), wcat.stride(0), Mk, G * C, K, L.ptr(Tsk), G * r,
               L.ptr(sites[0].b16), r, r, C, 0, r, None, None, 0, L.ptr(yk), G * C, None, 0, None, rps, int(row0), L.ptr(ws),
               ws.numel() * 4, L.stream_ptr())
        y = yk[M:] if twin else yk
        ctx.packs, ctx.sites, ctx.rps, ctx.wcatT = packs, sites, rps, wcatT
        ctx.ds_accum = getattr(S, "_aql_ds_accum", None)
        ctx.rep = rep[rep.shape[0] - S16.shape[0]:]      # the scale rows of the rows autograd sees
        ctx.save_for_backward(x2d, Tk[M:] if twin else Tk, Tsk[M:] if twin else Tsk, S16)
        return tuple(y[:, g * C:(g + 1) * C] for g in range(G))

    @staticmethod
    def backward(ctx, *dys):
        x2d, T, Ts, S16 = ctx.saved_tensors
        sites, packs, rps = ctx.sites, ctx.packs, ctx.rps
        G, r, C = len(sites), sites[0].rank, packs[0].N
        M, K = x2d.shape
        dev = x2d.device
        dcat = packed_columns(dys, M, C)      # the attention backward's [dY_1 | .. | dY_G], in place
        if dcat is None:
            dcat = torch.cat([torch.zeros(M, C, dtype=torch.bfloat16, device=dev) if d is None else d for d in dys], dim=1)
        dTs = torch.empty(M, G * r, dtype=torch.bfloat16, device=dev)
        dT = torch.empty_like(dTs)
        L.call("aql_gemm_bf16_grouped", L.ptr(dcat), G * C, L.ptr(sites[0].bt16), C, M, G * r, C, None, 0, None, 0, 0, r, C, 0, None,
               None, 0, L.ptr(dTs), G * r, L.ptr(dT), G * r, L.ptr(ctx.rep), rps, 0, None, 0, L.stream_ptr())
        dx = None
        if ctx.needs_input_grad[0]:
            at0 = sites[0].at16
            acatT = at0.as_strided((K, G * r), (G * r, 1), at0.storage_offset())
            dx = gemm_bf16(dcat, ctx.wcatT, None, dT, acatT)
        acc = ctx.ds_accum
        nb = S16.shape[0]
        dsg = getattr(acc, "_aql_dsg", None)
        if dsg is None:
            dsg = acc._aql_dsg = {}
        t = dsg.get(G)
        if t is None or t[0].shape != (nb, G * r) or t[0].device != dev:
            t = dsg[G] = [torch.zeros(nb, G * r, dtype=torch.float32, device=dev), False]
        t[1] = True
        if not DEFERRED.add_ds(dTs, T, t[0], nb, rps, G * r):
            L.call("aql_lora_ds", L.ptr(dTs), L.ptr(T), nb, rps, G * r, L.ptr(t[0]), L.stream_ptr())
        for g, site in enumerate(sites):
            DEFERRED.add_tn(dcat[:, g * C:(g + 1) * C], Ts[:, g * r:(g + 1) * r], site.gb)
            DEFERRED.add_tn(dT[:, g * r:(g + 1) * r], x2d, site.ga)
        return dx, None, None, None, None, None, None, None


def lora_linear_grouped_wide(x2d, wcat, wcatT, packs, sites, S, S16, rps):
    return GroupedWideFn.apply(x2d, wcat, wcatT, tuple(packs), tuple(sites), S, S16, rps)


def grouped_lora_ok(x2d, packs, sites, S16):
    """The one-launch grouped form applies: rank 32 everywhere, 160-column groups, stacked bf16 copies, fused kernel enabled."""
    if S16 is None or os.environ.get("AQL_LORA_FUSED", "1") == "0" or not _GROUPED or REF_ROUNDING:
        return False
    if len(sites) > 32 or any(s is None or s.rank != 32 for s in sites) or any(p.N % 160 != 0 or p.bias is not None for p in packs):
        return False
    if x2d.shape[1] % 8 != 0:
        return False
    return adjacent([s.a16 for s in sites]) and adjacent([s.b16 for s in sites])


def lora_linear_grouped(x2d, wcat, packs, sites, S, S16, rps):
    return GroupedLoraFn.apply(x2d, wcat, tuple(packs), tuple(sites), S, S16, rps)


def lora_linear(x2d, packed, site=None, S=None, S16=None, rps=1, residual=None, geglu=False):
    """S: the [nb, r] scale as seen by autograd (its gradient dS is returned in S.dtype, accumulated in fp32);
    S16: its bf16 copy read by the kernels.  ``geglu``: see LoraLinearFn."""
    want_h = torch.is_grad_enabled() and (x2d.requires_grad or (S is not None and torch.is_tensor(S) and S.requires_grad))
    return LoraLinearFn.apply(x2d, packed, site, S, S16, rps, residual, geglu, want_h)


# ------------------------------------------------------------------------- row-resident chains (csrc/aql_chain.hip)
CHAIN = os.environ.get("AQL_CHAIN", "1") != "0"   # A/B hook: 0 = every linear / LayerNorm of the transformer block as its own launch
CHAIN_BWD = True   # False = the chains' backward as separate launches (module attribute; measured neutral, kept for the launch count)
CHAIN_R320 = True   # False = rank-320 linears as aql_lora_down + aql_gemm_bf16 launches (module attribute)
CHAIN_MIN_TILES = 128   # below this many 128-row tiles the chip is mostly idle: unfused (module attribute: tests lower it)


class ChainStage:
    """One linear of a chain (static description).  ``keep``: the output tile stays in LDS as the next stage's input, after
    ``+ residual`` (use_res) and LayerNorm (ln = module with weight / bias / eps); emit_out / emit_n: the (pre-LayerNorm) output / the
    normalised rows are results of the chain (tensors autograd sees), not only saved state."""
    __slots__ = ("packed", "site", "keep", "use_res", "ln", "emit_out", "emit_n", "oscale")

    def __init__(self, packed, site, keep, use_res=False, ln=None, emit_out=True, emit_n=False, oscale=1.0):
        self.packed, self.site, self.keep, self.use_res, self.ln, self.emit_out, self.emit_n = packed, site, keep, use_res, ln, emit_out, emit_n
        # a DIRECT (keep = False) stage writes  out = bf16((x.W^T + LoRA + bias) * oscale):  attn1.to_q hands the attention kernels
        # q * (d^-1/2 log2 e) (attention(..., q_prescaled=True)); its backward still receives the gradient of the UNSCALED q
        self.oscale = float(oscale)


def chain_ok(x2d, stages, S16, rps):
    """The chain kernel takes these linears: 320 -> 320, all with the rank-32 (or all with the rank-320) LoRA (bf16 scale rows) or all
    without (the fused-weight / clean passes: S16 None), whole 64-row tiles, enough of them to fill half the chip."""
    if not CHAIN or REF_ROUNDING or x2d.dtype != torch.bfloat16 or x2d.dim() != 2 or x2d.stride(1) != 1:
        return False
    M = x2d.shape[0] * (2 if _full(x2d) is not None else 1)
    if x2d.shape[1] != 320 or M % 64 or rps % 64 or M // 64 < CHAIN_MIN_TILES or (M * x2d.stride(0) * 2) >= (1 << 30):
        return False
    if S16 is None and M // 64 < 2 * CHAIN_MIN_TILES:
        # LoRA-free passes at CFG batch 2 (8192 rows = 128 tiles of 64, half the chip): measured 5.56 vs 5.54 ms per guided forward
        # against the per-launch path, whose q | k | v is ONE 960-wide GEMM on 768 tiles (tools/time_infer_forward.py) -- not taken
        return False
    rank = None if S16 is None else S16.shape[1]
    if rank not in (None, 32, 320) or (rank == 320 and (not CHAIN_R320 or M // 64 < 2 * CHAIN_MIN_TILES)):
        return False       # (rank 320: one 150 KB workgroup per CU on 64-row tiles)
    for st in stages:
        if st.packed.N != 320 or st.packed.K != 320 or (S16 is None) != (st.site is None) or (st.site is not None and st.site.rank != rank):
            return False
        if rank == 320 and wside_ok(st.packed, st.site, S16, rps):
            return False   # the opt-in weight-side form owns these sites
    return os.environ.get("AQL_LORA_FUSED", "1") != "0"


def chain_input_ok(t, x2d):
    """A later chain of a block reads an attention output: it must have the geometry chain_ok accepted for the block's input x2d
    (bf16 rows of 320, unit column stride, under 1 GiB, a twin view exactly when x2d is one)."""
    return (t.dtype == torch.bfloat16 and t.dim() == 2 and t.shape == x2d.shape and t.stride(1) == 1
            and (_full(t) is not None) == (_full(x2d) is not None)
            and t.shape[0] * (2 if _full(t) is not None else 1) * t.stride(0) * 2 < (1 << 30))


class ChainFn(torch.autograd.Function):
    """A chain of 320 -> 320 LoRA linears with the row-local operations between them (bias, residual add, LayerNorm) as ONE launch
    (aql_lora_chain_fwd): attn.to_out + residual -> LayerNorm -> next projection(s) of BasicTransformerBlock.forward and
    proj_in -> norm1 -> to_q | to_k | to_v (scripts/lib/original_unet.py:786-806, 856-861; utils/lora_modules.py:9-26, 56-62).
    Forward: bit-identical to LoraLinearFn / LayerNormFn / GroupedLoraFn in sequence.  Backward: exactly their backward launches, in
    their order (the saved tensors are the same ones).  Outputs: per stage, in order, the emitted `out` then the emitted `n`."""

    @staticmethod
    def forward(ctx, x2d, res, S, S16, rps, stages):
        _req(x2d, "lora chain")
        M, C = x2d.shape
        dev = x2d.device
        xk = _full(x2d)
        twin = xk is not None
        if not twin:
            xk = x2d
        resk = None
        if res is not None:
            resk = _need_full(res, "the residual") if twin else res
        lora = S16 is not None
        rank = S16.shape[1] if lora else 32
        S16k = (_need_full(S16, "the LoRA scale") if twin else S16) if lora else None
        row0 = M if (twin and _TWIN_SKIP) else 0
        kst, outs, saved = [], [], [x2d, S16]
        meta = []
        for st in stages:
            d = dict(W=st.packed.w, ldw=st.packed.w.stride(0), bias=st.packed.bias, keep=int(st.keep), oscale=st.oscale)
            T = Ts = None
            if lora:
                Tk, T = _alloc((M, rank), torch.bfloat16, dev, twin)
                Tsk, Ts = _alloc((M, rank), torch.bfloat16, dev, twin)
                d.update(Ad=st.site.a16, Bup=st.site.b16, T=Tk, Ts=Tsk)
            ok = o = nk = n = stk = stt = None
            if st.emit_out or st.ln is not None or not st.keep:
                ok, o = _alloc((M, C), torch.bfloat16, dev, twin)
                d.update(out=ok, ldo=C)
            if st.use_res:
                d.update(res=resk, ldr=resk.stride(0))
            if st.ln is not None:
                nk, n = _alloc((M, C), torch.bfloat16, dev, twin)
                stk, stt = _alloc((M, 2), torch.float32, dev, twin)
                d.update(ln=1, gamma=st.ln.weight, beta=st.ln.bias, eps=float(st.ln.eps), stats=stk, nout=nk, ldn=C,
                         nout_row0=0 if st.emit_n else row0)
            kst.append(d)
            if st.emit_out:
                outs.append(o)
            if st.emit_n:
                outs.append(n)
            meta.append((len(saved), o is not None, n is not None))
            saved += [T, Ts] + ([o] if o is not None else []) + ([n, stt] if n is not None else [])
        chain_fwd(xk, xk.stride(0), xk.shape[0], rps, row0, S16k, kst, rank)
        if not lora:   # the LoRA-free chains run under no_grad only (the clean pass / sampling with fused weights): nothing to save
            return tuple(outs)
        ctx.stages, ctx.meta, ctx.rps = stages, meta, rps
        ctx.ds_accum = getattr(S, "_aql_ds_accum", None)
        ctx.s_dtype = S.dtype
        # rank > 32: the q | k | v stages' backward as two grouped launches (wide_grouped_backward) needs the scale rows three times over
        ctx.rep3 = None
        if rank > 32 and GROUPED_WIDE:
            rp = _rep_g(S, S16k, 3)
            ctx.rep3 = rp[rp.shape[0] - S16.shape[0]:]
        ctx.save_for_backward(*saved)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        sv = ctx.saved_tensors
        x2d, S16 = sv[0], sv[1]
        stages, rps = ctx.stages, ctx.rps
        ret_ds = ctx.needs_input_grad[2]
        # gradients of the emitted outputs, per stage
        gi = iter(grads)
        d_out, d_n = [], []
        for st in stages:
            d_out.append(next(gi) if st.emit_out else None)
            d_n.append(next(gi) if st.emit_n else None)
        # the tensors each stage saved, and the input each stage read (x2d, or the tile the previous keep stage left)
        T, Ts, O, N, ST, X = [], [], [], [], [], []
        cur = x2d
        for st, (p, has_o, has_n) in zip(stages, ctx.meta):
            T.append(sv[p])
            Ts.append(sv[p + 1])
            q = p + 2
            O.append(sv[q] if has_o else None)
            q += 1 if has_o else 0
            N.append(sv[q] if has_n else None)
            ST.append(sv[q + 1] if has_n else None)
            X.append(cur)
            if st.keep:
                cur = N[-1] if has_n else O[-1]
        dR, d_res, dS_sum = None, None, None

        def add(a, b):
            return b if a is None else (a if b is None else a + b)

        def queue_site(k, dy_k, dTs_k, dT_k):     # dS and the weight gradients of linear k, as _grouped_backward queues them
            site = stages[k].site
            nb = S16.shape[0]
            if not DEFERRED.add_ds(dTs_k, T[k], ctx.ds_accum, nb, rps, 32):
                L.call("aql_lora_ds", L.ptr(dTs_k), L.ptr(T[k]), nb, rps, 32, L.ptr(ctx.ds_accum), L.stream_ptr())
            if not DEFERRED.add_tn(dy_k, Ts[k], site.gb):
                gemm_tn_acc(dy_k, Ts[k], site.gb)
            if not DEFERRED.add_tn(dT_k, X[k], site.ga):
                gemm_tn_acc(dT_k, X[k], site.ga)

        def bwd_stage(k, dTs_k, dT_k, **kw):
            st_ = stages[k]
            return dict(Wt=st_.packed.wt, ldw=st_.packed.wt.stride(0), BupT=st_.site.bt16, AT=st_.site.at16, dTs=dTs_k, dT=dT_k, **kw)

        M = x2d.shape[0]
        fused_bwd = (CHAIN_BWD and ctx.ds_accum is not None and DEFERRED is not None and M % 64 == 0 and rps % 64 == 0
                     and M // 64 >= CHAIN_MIN_TILES and S16.shape[1] == 32)      # (dS goes to the trainer's accumulator, as in _grouped_backward's one-launch branch)
        mk = lambda *shape: torch.empty(*shape, dtype=torch.bfloat16, device=x2d.device)   # noqa: E731

        g = len(stages) - 1
        while g >= 0:
            st = stages[g]
            # ---- backward chains (aql_lora_chain_bwd)
            if (fused_bwd and not st.keep and g >= 1 and g + 1 == len(stages) and dR is None and d_out[g] is not None
                    and stages[g - 1].keep and stages[g - 1].ln is not None and d_n[g - 1] is None
                    and (g - 1 > 0 or ctx.needs_input_grad[0])):
                # the trailing DIRECT linear's backward -> LayerNorm backward -> the keep linear's backward as ONE launch
                # (attn2.to_q backward -> norm2 backward -> attn1.to_out backward: the mirror of chain `a`)
                k, g = g, g - 1
                st = stages[g]
                ln_entry = dict(x=O[g], ldx=O[g].stride(0), stats=ST[g], gamma=st.ln.weight)
                dy_k = d_out[k].contiguous()
                dres = None if d_out[g] is None else d_out[g].contiguous()
                dhs, dx = mk(M, 320), mk(M, 320)
                dTs_k, dT_k, dTs_g, dT_g = mk(M, 32), mk(M, 32), mk(M, 32), mk(M, 32)
                chain_bwd(dy_k, dy_k.stride(0), M, rps, S16,
                          [bwd_stage(k, dTs_k, dT_k, keep=1), bwd_stage(g, dTs_g, dT_g, dX=dx, lddx=320, keep=0)],
                          [None, dict(ln_entry, dres=dres, lddres=0 if dres is None else dres.stride(0), out=dhs, ldo=320), None])
                queue_site(k, dy_k, dTs_k, dT_k)
                queue_site(g, dhs, dTs_g, dT_g)
                if st.use_res:
                    d_res = dhs
                dR = dx
                g -= 1
                continue
            if fused_bwd and st.keep and st.ln is not None and (g > 0 or ctx.needs_input_grad[0]):
                ln_entry = dict(x=O[g], ldx=O[g].stride(0), stats=ST[g], gamma=st.ln.weight)
                dn = add(dR, d_n[g])
                if dn is not None:
                    # LayerNorm backward -> this linear's backward (norm3 -> attn2.to_out; norm1 -> proj_in)
                    dn = dn.contiguous()
                    dres = None if d_out[g] is None else d_out[g].contiguous()
                    dhs, dx = mk(M, 320), mk(M, 320)
                    dTs_g, dT_g = mk(M, 32), mk(M, 32)
                    chain_bwd(dn, dn.stride(0), M, rps, S16, [bwd_stage(g, dTs_g, dT_g, dX=dx, lddx=320, keep=0)],
                              [dict(ln_entry, dres=dres, lddres=0 if dres is None else dres.stride(0), out=dhs, ldo=320), None])
                    queue_site(g, dhs, dTs_g, dT_g)
                    if st.use_res:
                        d_res = dhs
                    dR = dx
                    g -= 1
                    continue
            if not st.keep:
                lo_ = g
                while lo_ > 0 and not stages[lo_ - 1].keep:
                    lo_ -= 1
                run = list(range(lo_, g + 1))
                want_dx = lo_ > 0 or ctx.needs_input_grad[0]
                dx, dS = _grouped_backward([d_out[k] for k in run], X[lo_], [T[k] for k in run], [Ts[k] for k in run], S16,
                                           [stages[k].packed for k in run], [stages[k].site for k in run], rps, ctx.ds_accum,
                                           want_dx, ret_ds, ctx.s_dtype, rep=ctx.rep3)
                dR = add(dR, dx)
                dS_sum = add(dS_sum, dS)
                g = lo_ - 1
                continue
            dn = add(dR, d_n[g])
            if st.ln is not None:
                hs = O[g]
                dres = None if d_out[g] is None else d_out[g].contiguous()
                if dn is None:
                    dhs = dres
                else:
                    dhs = torch.empty_like(hs)
                    L.call("aql_layernorm_bwd", L.ptr(hs), L.ptr(dn.contiguous()), hs.shape[0], hs.shape[1], L.ptr(st.ln.weight),
                           L.ptr(ST[g]), L.ptr(dres), L.ptr(dhs), L.stream_ptr())
            else:
                dhs = add(dn, d_out[g])
            if st.use_res:
                d_res = dhs
            want_dx = g > 0 or ctx.needs_input_grad[0]
            dx, dS = _lora_backward(dhs.contiguous(), X[g], T[g], Ts[g], S16, st.packed, st.site, rps, ctx.ds_accum, want_dx, ret_ds,
                                    ctx.s_dtype, None)
            dR = dx
            dS_sum = add(dS_sum, dS)
            g -= 1
        return dR, d_res, dS_sum, None, None, None


def lora_chain(x2d, res, S, S16, rps, stages):
    return ChainFn.apply(x2d, res, S, S16, rps, tuple(stages))


# --------------------------------------------------------------------------------------------- conv 3x3
class Conv3x3Fn(torch.autograd.Function):
    """3x3 conv, pad 1, stride 1|2, optional nearest x2 upsample folded into the gather; channels-last bf16.
    Optional per-sample row bias (the ResNet time-embedding add) and residual fused into the epilogue."""

    @staticmethod
    def forward(ctx, x, packed, upsample, rowbias, residual, gn_next=False, gn_input=False):
        """gn_next: the ONLY consumer of the output is a GroupNorm called right after (ResnetBlock2D: conv1 -> norm2): a split-K
        launch leaves its finalize to that GroupNorm's first pass.  gn_input: the input is a GroupNorm's output read by this
        convolution only: the backward-data launch leaves ITS finalize to that GroupNorm's backward kernel."""
        _req(x, "conv3x3")
        cpad = x.shape[1] != packed.Cin and is_cpad(x, packed.Cin)
        if not cpad:
            x = as_cl(x)
        B, C, H, W = x.shape
        xk = _full(x)
        twin = xk is not None
        if not twin:
            xk = x
        Bk = xk.shape[0]
        if cpad:   # already packed.Cin channels wide in memory, zero beyond C
            xk = xk.as_strided((Bk, packed.Cin, H, W), xk.stride(), xk.storage_offset())
        elif C != packed.Cin:  # conv_in: zero-pad channels to the packed width
            xp = xk.new_zeros((Bk, packed.Cin, H, W)).contiguous(memory_format=CL)
            xp[:, :C] = xk
            xk = xp
        Hl, Wl = (H * 2, W * 2) if upsample else (H, W)
        Ho = (Hl + 2 - 3) // packed.stride + 1
        Wo = (Wl + 2 - 3) // packed.stride + 1
        yk, y = _alloc((B, packed.Cout, Ho, Wo), torch.bfloat16, x.device, twin, cl=True)
        ws = workspace(x.device)
        res = None if residual is None else as_cl(residual)
        resk = res if (res is None or not twin) else _need_full(res, "the conv residual")
        if twin and rowbias is not None and rowbias.shape[0] != Bk:
            raise L.AqlError("twin batch: the per-sample row bias must cover both halves")
        # maps above 1 GiB (the VAE's 256-channel 512 x 512 level at batch 16: 2.1 GiB) go through the kernel in sample chunks
        chunks = span_chunks(Bk, max(H * W * packed.Cin, Ho * Wo * packed.Cout) * 2)
        if DEFER_FINALIZE and gn_next and len(chunks) == 1 and packed.Cout_real == packed.Cout:
            import ctypes
            ns = ctypes.c_int(1)
            ws = defer_workspace(x.device)
            L.call("aql_conv3x3_fwd_defer", L.ptr(xk), Bk, H, W, packed.Cin, L.ptr(packed.wk), L.ptr(packed.bias), packed.Cout,
                   packed.stride, int(upsample), L.ptr(rowbias), 0 if rowbias is None else rowbias.stride(0), L.ptr(resk),
                   L.ptr(yk), L.ptr(ws), ws.numel() * 4, ctypes.byref(ns), L.stream_ptr())
            if ns.value > 1:
                p = _Pending()
                p.t, p.ws, p.splits, p.M, p.N = yk, ws, ns.value, Bk * Ho * Wo, packed.Cout
                p.bias, p.rowbias, p.rowbias_ld = packed.bias, rowbias, (0 if rowbias is None else rowbias.stride(0))
                p.rps, p.residual = Ho * Wo, resk
                _PENDING[_skey(x.device)] = p
        else:
            for b0, nb in chunks:
                L.call("aql_conv3x3_fwd", L.ptr(xk[b0:b0 + nb]), nb, H, W, packed.Cin, L.ptr(packed.wk), L.ptr(packed.bias), packed.Cout,
                       packed.stride, int(upsample), L.ptr(None if rowbias is None else rowbias[b0:]),
                       0 if rowbias is None else rowbias.stride(0), L.ptr(None if resk is None else resk[b0:b0 + nb]),
                       L.ptr(yk[b0:b0 + nb]), L.ptr(ws), ws.numel() * 4, L.stream_ptr())
        ctx.defer_bwd = bool(gn_input)
        ctx.packed, ctx.upsample, ctx.in_shape, ctx.c_in = packed, upsample, (B, H, W), C
        ctx.has_rb, ctx.has_res = rowbias is not None, residual is not None
        if packed.Cout_real != packed.Cout:
            y = y[:, :packed.Cout_real]
        return y

    @staticmethod
    def backward(ctx, dy):
        packed = ctx.packed
        B, H, W = ctx.in_shape
        if packed.Cout_real != packed.Cout:
            dyp = dy.new_zeros((B, packed.Cout, dy.shape[2], dy.shape[3])).contiguous(memory_format=CL)
            dyp[:, :packed.Cout_real] = dy
            dy = dyp
        dy = as_cl(dy)
        dx = None
        if ctx.needs_input_grad[0]:
            Hl, Wl = (H * 2, W * 2) if ctx.upsample else (H, W)
            ws = workspace(dy.device)
            du = torch.empty((B, packed.Cin, Hl, Wl), dtype=torch.bfloat16, device=dy.device, memory_format=CL)
            chunks = span_chunks(B, max(Hl * Wl * packed.Cin, dy.shape[2] * dy.shape[3] * packed.Cout) * 2)
            if DEFER_FINALIZE and ctx.defer_bwd and len(chunks) == 1 and not ctx.upsample and ctx.c_in == packed.Cin:
                import ctypes
                ns = ctypes.c_int(1)
                ws = defer_workspace(dy.device)
                L.call("aql_conv3x3_bwd_data_defer", L.ptr(dy), B, Hl, Wl, packed.Cin, L.ptr(packed.wt), packed.Cout, packed.stride,
                       L.ptr(du), L.ptr(ws), ws.numel() * 4, ctypes.byref(ns), L.stream_ptr())
                if ns.value > 1:   # the GroupNorm backward in front of this convolution sums the slabs (GroupNormSiluFn.backward)
                    p = _Pending()
                    p.t, p.ws, p.splits, p.M, p.N = du, ws, ns.value, B * Hl * Wl, packed.Cin
                    p.bias = p.rowbias = p.residual = None
                    p.rowbias_ld, p.rps = 0, 1
                    _PENDING[_skey(dy.device)] = p
            else:
                for b0, nb in chunks:
                    L.call("aql_conv3x3_bwd_data", L.ptr(dy[b0:b0 + nb]), nb, Hl, Wl, packed.Cin, L.ptr(packed.wt), packed.Cout,
                           packed.stride, L.ptr(du[b0:b0 + nb]), L.ptr(ws), ws.numel() * 4, L.stream_ptr())
            if ctx.upsample:
                dx = torch.empty((B, packed.Cin, H, W), dtype=torch.bfloat16, device=dy.device, memory_format=CL)
                L.call("aql_upsample2x_bwd", L.ptr(du), B, H, W, packed.Cin, L.ptr(dx), L.stream_ptr())
            else:
                dx = du
            if ctx.c_in != packed.Cin:
                dx = dx[:, :ctx.c_in]
        drb = None
        if ctx.has_rb and ctx.needs_input_grad[3]:
            drb = dy.float().sum(dim=(2, 3)).to(torch.bfloat16)
        return dx, None, None, drb, (dy if ctx.has_res else None), None, None


def conv3x3(x, packed, upsample=False, rowbias=None, residual=None, gn_next=False, gn_input=False):
    return Conv3x3Fn.apply(x, packed, upsample, rowbias, residual, gn_next, gn_input)


# ------------------------------------------------------------------------------ skip-connection concat
_CAT_FUSED = True   # False = two strided torch copies forward, slice views backward (module attribute)


class CatChannelsFn(torch.autograd.Function):
    """torch.cat([a, b], dim=1) on channels-last maps (the up-block skip connections, original_unet.py:1133,1224), twin-batch
    aware: the concatenation is built for both halves, autograd sees the second half."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = as_cl(a), as_cl(b)
        B, Ca, H, W = a.shape
        Cb = b.shape[1]
        ak = _full(a)
        twin = ak is not None
        bk = _need_full(b, "the skip connection") if twin else b
        if not twin:
            ak = a
        outk, out = _alloc((B, Ca + Cb, H, W), a.dtype, a.device, twin, cl=True)
        if _CAT_FUSED and a.dtype == torch.bfloat16 and Ca % 8 == 0 and Cb % 8 == 0 and ak.is_contiguous(memory_format=CL) \
                and bk.is_contiguous(memory_format=CL):
            L.call("aql_cat_channels", L.ptr(ak), L.ptr(bk), outk.shape[0] * H * W, Ca, Cb, L.ptr(outk), L.stream_ptr())
        else:
            outk[:, :Ca] = ak
            outk[:, Ca:] = bk
        ctx.ca = Ca
        return out

    @staticmethod
    def backward(ctx, dy):
        Ca = ctx.ca
        B, C, H, W = dy.shape
        Cb = C - Ca
        if _CAT_FUSED and dy.dtype == torch.bfloat16 and Ca % 8 == 0 and Cb % 8 == 0 and dy.is_contiguous(memory_format=CL):
            # both slices dense in one launch: their consumers (conv / GEMM backward) need dense operands, and the gradient
            # accumulation of the skip tensor then adds two dense tensors
            da = torch.empty((B, Ca, H, W), dtype=dy.dtype, device=dy.device, memory_format=CL)
            db = torch.empty((B, Cb, H, W), dtype=dy.dtype, device=dy.device, memory_format=CL)
            L.call("aql_split_channels", L.ptr(dy), B * H * W, Ca, Cb, L.ptr(da), L.ptr(db), L.stream_ptr())
            return da, db
        return dy[:, :Ca], dy[:, Ca:]


def cat_channels(a, b):
    return CatChannelsFn.apply(a, b)


# ------------------------------------------------------------------------------------------- norms
class GroupNormSiluFn(torch.autograd.Function):
    """GroupNorm(32)+SiLU.  With ``passthrough`` the input is ALSO returned (as a view) so that the residual / shortcut
    branch hangs off this node: backward then receives both gradients at once and adds the residual one inside the
    GroupNorm backward kernel instead of a separate elementwise-add launch."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, silu, passthrough=False):
        _req(x, "groupnorm")
        x = as_cl(x)
        B, C, H, W = x.shape
        xk = _full(x)
        twin = xk is not None
        if not twin:
            xk = x
        Bk = xk.shape[0]
        yk, y = _alloc((B, C, H, W), x.dtype, x.device, twin, cl=True)
        statsk, stats = _alloc((B, 32, 2), torch.float32, x.device, twin)
        p = take_pending(xk) if _PENDING else None
        if p is not None:   # xk is still the fp32 slabs of the split-K convolution in front: this launch finishes it too
            rc = L.call_raw("aql_groupnorm_silu_fwd_slabs", L.ptr(p.ws), p.splits, L.ptr(p.bias), L.ptr(p.rowbias), int(p.rowbias_ld),
                            L.ptr(p.residual), L.ptr(xk), Bk, H * W, C, L.ptr(gamma), L.ptr(beta), float(eps), int(silu), L.ptr(yk),
                            L.ptr(statsk), L.stream_ptr())
            if rc == 100:   # a map the one-launch GroupNorm does not take: the finalize launch after all
                p.finalize()
            else:
                L.check(rc, "aql_groupnorm_silu_fwd_slabs")
        if p is None or rc == 100:
            L.call("aql_groupnorm_silu_fwd", L.ptr(xk), Bk, H * W, C, L.ptr(gamma), L.ptr(beta), float(eps), int(silu),
                   L.ptr(yk), L.ptr(statsk), L.ptr(_gn_scratch(x.device, Bk)), L.stream_ptr())
        ctx.save_for_backward(x, gamma, beta, stats)
        ctx.silu = silu
        if passthrough:
            return y, x.view_as(x)
        return y

    @staticmethod
    def backward(ctx, dy, dres=None):
        x, gamma, beta, stats = ctx.saved_tensors
        B, C, H, W = x.shape
        if dy is None:  # only the pass-through branch carried a gradient
            return (None if dres is None else as_cl(dres)), None, None, None, None, None
        dy = as_cl(dy)
        dres = None if dres is None else as_cl(dres)
        dx = torch.empty_like(x, memory_format=CL)
        p = take_pending(dy) if _PENDING else None
        rc = 100
        if p is not None:   # dy is still the fp32 slabs of the backward-data convolution behind: consumed here, never written
            rc = L.call_raw("aql_groupnorm_silu_bwd_slabs", L.ptr(x), L.ptr(p.ws), p.splits, B, H * W, C, L.ptr(gamma), L.ptr(beta),
                            int(ctx.silu), L.ptr(stats), L.ptr(dres), L.ptr(dx), L.stream_ptr())
            if rc == 100:
                p.finalize()
            else:
                L.check(rc, "aql_groupnorm_silu_bwd_slabs")
        if rc == 100:
            L.call("aql_groupnorm_silu_bwd", L.ptr(x), L.ptr(dy), B, H * W, C, L.ptr(gamma), L.ptr(beta), int(ctx.silu),
                   L.ptr(stats), L.ptr(dres), L.ptr(dx), L.ptr(_gn_scratch(x.device, B)), L.stream_ptr())
        return dx, None, None, None, None, None


def groupnorm_silu(x, gamma, beta, eps, silu):
    return GroupNormSiluFn.apply(x, gamma, beta, eps, silu)


def groupnorm_silu_res(x, gamma, beta, eps, silu):
    """-> (normalised, x) with the residual gradient folded into the GroupNorm backward kernel."""
    return GroupNormSiluFn.apply(x, gamma, beta, eps, silu, True)


class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x2d, gamma, beta, eps, passthrough=False):
        _req(x2d, "layernorm")
        M, C = x2d.shape
        xk = _full(x2d)
        twin = xk is not None
        if not twin:
            xk = x2d
        yk, y = _alloc((M, C), x2d.dtype, x2d.device, twin)
        statsk, stats = _alloc((M, 2), torch.float32, x2d.device, twin)
        L.call("aql_layernorm_fwd", L.ptr(xk), xk.shape[0], C, L.ptr(gamma), L.ptr(beta), float(eps), L.ptr(yk), L.ptr(statsk),
               L.stream_ptr())
        ctx.save_for_backward(x2d, gamma, stats)
        if passthrough:
            return y, x2d.view_as(x2d)
        return y

    @staticmethod
    def backward(ctx, dy, dres=None):
        x2d, gamma, stats = ctx.saved_tensors
        if dy is None:
            return (None if dres is None else dres.contiguous()), None, None, None, None
        dy = dy.contiguous()
        dres = None if dres is None else dres.contiguous()
        M, C = x2d.shape
        dx = torch.empty_like(x2d)
        L.call("aql_layernorm_bwd", L.ptr(x2d), L.ptr(dy), M, C, L.ptr(gamma), L.ptr(stats), L.ptr(dres), L.ptr(dx),
               L.stream_ptr())
        return dx, None, None, None, None


def layernorm(x2d, gamma, beta, eps=1e-5):
    return LayerNormFn.apply(x2d, gamma, beta, eps)


def layernorm_res(x2d, gamma, beta, eps=1e-5):
    """-> (normalised, x2d) with the residual gradient folded into the LayerNorm backward kernel."""
    return LayerNormFn.apply(x2d, gamma, beta, eps, True)


class GegluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h2d):
        _req(h2d, "geglu")
        M, F2 = h2d.shape
        hk = _full(h2d)
        twin = hk is not None
        if not twin:
            hk = h2d
        outk, out = _alloc((M, F2 // 2), torch.bfloat16, h2d.device, twin)
        L.call("aql_geglu_fwd", L.ptr(hk), hk.shape[0], F2 // 2, L.ptr(outk), L.stream_ptr())
        ctx.save_for_backward(h2d)
        return out

    @staticmethod
    def backward(ctx, dy):
        (h2d,) = ctx.saved_tensors
        dy = dy.contiguous()
        M, F2 = h2d.shape
        din = torch.empty_like(h2d)
        L.call("aql_geglu_bwd", L.ptr(h2d), L.ptr(dy), M, F2 // 2, L.ptr(din), L.stream_ptr())
        return din


def geglu(h2d):
    return GegluFn.apply(h2d)


# --------------------------------------------------------------------------------------- attention
class AttentionFn(torch.autograd.Function):
    """softmax(Q K^T / sqrt(d)) V over heads packed along the channel axis: q [B,Nq,H*d], k/v [B,Nk,H*d]."""

    @staticmethod
    def forward(ctx, q, k, v, heads, q_prescaled=False, pack_grads=False):
        """pack_grads (the projections came from ops.GroupedWideFn): 1 = backward writes dq | dk | dv as the column blocks of ONE
        [B, N, 3C] buffer, 2 = dk | dv as the column blocks of one [B, Nk, 2C] buffer (aql_sdpa_bwd_ex), which the grouped backward of
        the projections reads in place."""
        _req(q, "attention")
        B, Nq, C = q.shape
        Nk = k.shape[1]
        d = C // heads
        qk = _full(q)
        twin = qk is not None
        if twin:
            kk, vk = _need_full(k, "attention k"), _need_full(v, "attention v")
        else:
            qk, kk, vk = q, k, v
        ok, o = _alloc((B, Nq, C), q.dtype, q.device, twin)   # q may be a strided view of a packed q|k|v GEMM output
        lsek, lse = _alloc((B, heads, Nq), torch.float32, q.device, twin)
        # q_prescaled: q was multiplied by d^-1/2 log2(e) by its producer (ChainStage.oscale) -- aql_sdpa_*_qpre
        L.call("aql_sdpa_fwd_qpre" if q_prescaled else "aql_sdpa_fwd", L.ptr(qk), qk.stride(1), L.ptr(kk), kk.stride(1), L.ptr(vk), vk.stride(1),
               qk.shape[0], heads, Nq, Nk, d, float(d ** -0.5), L.ptr(ok), ok.stride(1), L.ptr(lsek), L.stream_ptr())
        ctx.save_for_backward(q, k, v, o, lse)
        ctx.heads, ctx.qpre = heads, bool(q_prescaled)
        ctx.pack = int(pack_grads) if (int(pack_grads) == 2 or Nq == Nk) else 0      # 1: [dq | dk | dv], 2: [dk | dv]
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse = ctx.saved_tensors
        do = do.contiguous()
        B, Nq, C = q.shape
        Nk = k.shape[1]
        heads = ctx.heads
        d = C // heads
        # dense gradients whatever the strides of q / k / v (they may be column views of a grouped projection's output)
        delta = torch.empty(B, heads, Nq, dtype=torch.float32, device=q.device)
        ws = workspace(q.device)   # split-Q partials of dK/dV when Nk is short (cross-attention)
        if ctx.pack:
            if ctx.pack == 1:
                dqkv = torch.empty(B, Nq, 3 * C, dtype=q.dtype, device=q.device)
                dq, dk, dv = dqkv[..., :C], dqkv[..., C:2 * C], dqkv[..., 2 * C:]
                ldgq = ldgkv = 3 * C
            else:
                dq = torch.empty(B, Nq, C, dtype=q.dtype, device=q.device)
                dkv = torch.empty(B, Nk, 2 * C, dtype=q.dtype, device=q.device)
                dk, dv = dkv[..., :C], dkv[..., C:]
                ldgq, ldgkv = 0, 2 * C
            L.call("aql_sdpa_bwd_ex", int(ctx.qpre), L.ptr(q), q.stride(1), L.ptr(k), k.stride(1), L.ptr(v), v.stride(1), L.ptr(o),
                   L.ptr(do), o.stride(1), L.ptr(lse), L.ptr(delta), B, heads, Nq, Nk, d, float(d ** -0.5),
                   L.ptr(dq), L.ptr(dk), L.ptr(dv), ldgq, ldgkv, L.ptr(ws), ws.numel() * 4, L.stream_ptr())
            return dq, dk, dv, None, None, None
        dq, dk, dv = (torch.empty(t.shape, dtype=t.dtype, device=t.device) for t in (q, k, v))
        L.call("aql_sdpa_bwd_qpre" if ctx.qpre else "aql_sdpa_bwd", L.ptr(q), q.stride(1), L.ptr(k), k.stride(1), L.ptr(v), v.stride(1), L.ptr(o),
               L.ptr(do), o.stride(1), L.ptr(lse), L.ptr(delta), B, heads, Nq, Nk, d, float(d ** -0.5),
               L.ptr(dq), L.ptr(dk), L.ptr(dv), L.ptr(ws), ws.numel() * 4, L.stream_ptr())
        return dq, dk, dv, None, None, None


QPRE = os.environ.get("AQL_QPRE", "1") != "0"   # A/B hook: 0 = attn1.to_q of the chains unscaled, attention on aql_sdpa_fwd / _bwd


def attention(q, k, v, heads, q_prescaled=False, pack_grads=False):
    return AttentionFn.apply(q, k, v, heads, q_prescaled, pack_grads)


# ------------------------------------------------------------------------------------------- loss
class MseFn(torch.autograd.Function):
    """F.mse_loss(pred.float(), target.float(), reduction="mean") (ppft_train.py:1051): loss and d(pred) in one
    pass over the two bf16 tensors."""

    @staticmethod
    def forward(ctx, pred, target, unit_grad=False):
        _req(pred, "mse")
        ctx.unit_grad = unit_grad
        p = pred.contiguous() if pred.is_contiguous() else as_cl(pred)
        t = target.contiguous() if pred.is_contiguous() else as_cl(target)
        loss = torch.empty(1, dtype=torch.float32, device=pred.device)
        dpred = torch.empty_like(p)
        L.call("aql_mse_fwd_bwd", L.ptr(p), L.ptr(t), p.numel(), L.ptr(loss), L.ptr(dpred), L.stream_ptr())
        ctx.save_for_backward(dpred)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dpred,) = ctx.saved_tensors
        if ctx.unit_grad:   # the caller promises loss.backward() with the implicit seed of 1.0: no cast + multiply launches
            return dpred, None, None
        return dpred * g.to(dpred.dtype), None, None


def mse_loss(pred, target, unit_grad=False):
    return MseFn.apply(pred, target, unit_grad)
