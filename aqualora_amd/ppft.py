"""Prior-Preserving Fine-Tuning step (reference train/ppft_train.py:987-1068) on one MI355X per process.

    S      = mapper(m)                                   :990
    wm     = sec_encoder(m) * 0.18215      (no grad)     :994-996
    x_t    = add_noise(z, eps, t);  x_t_wm = add_noise(z + wm, eps, t)     :1010-1011
    clean  = unet(x_t,    t, ctx, scale = 0).detach()    :1026-1029   (LoRA branch skipped: identical to scale 0)
    pred   = unet(x_t_wm, t, ctx, scale = S)             :1032-1035
    loss   = mse(pred, clean)                            :1051
    backward; all-reduce(mean) of LoRA + mapper grads    :1058        (RCCL over xGMI, one flat buffer)
    clip_grad_norm_(LoRA params, 1.0); AdamW; lr step    :1059-1068   (mapper is not clipped, like the reference)

Inputs are injected (z = already-scaled VAE latents, ctx = text-encoder states): the frozen VAE / CLIP encoders are
outside this path.  All LoRA + mapper parameters, gradients and AdamW moments live in one flat
fp32 buffer (lora.LoraBank), so the exchange is a single collective and the optimizer two kernel launches.
"""
import os

import torch
import torch.distributed as dist

from . import _lib as L
from . import dp, ops
from .lora import LoraBank, inject_lora, patch_lora_forwards
from .watermark import customDDPMScheduler

_EARLY_DW = False  # True = the split (early / late) weight-gradient launches of the data-parallel form on a single GPU (measured neutral)
_PROLOGUE = True   # False = the generic (15-launch) head of the twin step (module attribute; the environment hook left in round 6)

VAE_SCALING = 0.18215


def _feed(static, z, msg, eps, t, ctx):
    """Fresh inputs -> the captured step's static buffers.  One multi-tensor copy per dtype (torch._foreach_copy_) instead of five
    4-5 us launches in front of every replay; falls back to per-tensor copies where the foreach form is not available."""
    dst, src = [], []
    for k, v in (("z", z), ("msg", msg), ("eps", eps), ("t", t), ("ctx", ctx)):
        if v is not static[k]:
            dst.append(static[k])
            src.append(v)
    if not dst:
        return
    try:
        torch._foreach_copy_(dst, src)
    except (AttributeError, RuntimeError):
        for d, v in zip(dst, src):
            d.copy_(v)


class PPFTTrainer:
    def __init__(self, unet, mapper, sec_encoder, rank, learning_rate=1e-4, adam_beta1=0.9, adam_beta2=0.999,
                 adam_weight_decay=1e-2, adam_epsilon=1e-8, max_grad_norm=1.0, lr_lambda=None, lora_state=None,
                 process_group=None, micro_batches=1, twin=None):
        dev = unet.device
        if dev.type != "cuda":
            raise L.AqlError("PPFTTrainer needs the U-Net on an MI355X (cuda device); there is no CPU path")
        self.unet, self.mapper, self.sec_encoder = unet, mapper, sec_encoder
        self.keys = None
        if not any(getattr(m, "lora_layer", None) is not None for m in unet.modules()):
            inject_lora(unet, rank, lora_state=lora_state)
        patch_lora_forwards(unet)
        self.mapper.to(dev)
        self.sec_encoder.to(dev)
        self.bank = LoraBank(unet, extra_params=[mapper.bit_embeddings.weight])
        self.pg = process_group
        # Our own RCCL communicator (aql_comm_*, include/aqualora_hip.h): collectives on a stream of OUR choice, capturable
        # into the step graph.  None on a single GPU, under a non-RCCL backend (gloo CPU tests), without AQL_COMM=1 (opt-in), or when its
        # self-test fails on this box -- then the torch.distributed exchange below is used (`comm_note` says which and why).
        self.comm, self.comm_note = dp.make_comm(process_group)
        # the overlapped exchange runs ONE backward pass per step: micro-batched steps use the bucketed torch.distributed exchange
        self.overlap = self.comm is not None and max(1, micro_batches) == 1
        # DDP broadcasts rank 0's parameters when it wraps the trainable modules (accelerator.prepare,
        # ppft_train.py:905-912).  inject_lora draws N(0, 1/r) and MapperNet an orthogonal table from each process's own
        # RNG (the reference default is seed=None): without this sync every rank would train a different replica on
        # averaged gradients.  All trainable state (LoRA + mapper) lives in the one flat buffer.
        if dp.world_size(process_group) > 1:
            if self.overlap:
                self.comm.broadcast_(self.bank.flat, root=0)
            else:
                dp.broadcast_(self.bank.flat, process_group, src=0)
            self.bank.refresh()
        self.scheduler = customDDPMScheduler(device=dev)
        self.hp = (adam_beta1, adam_beta2, adam_epsilon, adam_weight_decay)
        self.max_grad_norm = max_grad_norm
        self.base_lr = learning_rate
        self.lr_lambda = lr_lambda or (lambda s: 1.0)
        self.global_step = 0
        self.lr_t = torch.full((1,), learning_rate * self.lr_lambda(0), dtype=torch.float32, device=dev)
        self.step_t = torch.zeros(1, dtype=torch.int32, device=dev)
        self.sumsq = torch.zeros(1, dtype=torch.float32, device=dev)
        self.ds_accum = None
        self.micro = micro_batches
        # twin batch: clean + watermarked forward as one 2B-sample pass (ops._Dual); AQL_TWIN=0 restores the two-stream form
        import os
        self.twin = (os.environ.get("AQL_TWIN", "1") != "0") if twin is None else bool(twin)
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(2 * max(1, micro_batches))]
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        # data parallel: weight-gradient GEMMs run in buckets, each followed by its slice of the all-reduce (see
        # exchange_bucketed); single GPU: one grouped launch at the end of backward, no collective
        self.bucketed = (not self.overlap and dp.exchange_active(process_group)
                         and dp.bucket_count(4 * self.bank.n_lora) > 1)
        self.reducer = dp.BucketedAllreduce(process_group)
        # `split`: the weight-gradient GEMMs of the up path are launched from the backward hook on the mid-block output, so that
        # the all-reduce of their region (forked onto the side stream) runs under the mid / down backward.  The GEMMs themselves
        # stay on the main stream: on a side stream, concurrent with backward, they cost the step +0.53 ms (23.33 -> 23.86 ms, A/B
        # on one MI355X) -- the backward kernels lose more to the contention than the HBM-bound launch hides.  Single GPU: off
        # (nothing to overlap; AQL_EARLY_DW=1 forces the split form for tests / A-B).
        self.split = self.overlap or (not dp.exchange_active(process_group) and max(1, micro_batches) == 1
                                      and _EARLY_DW)
        self.side = torch.cuda.Stream(device=dev) if self.split else None   # forked / joined inside the step (and its graph)
        # backward legs that end with a hook-driven exchange (lora.backward_stage): 3 = up path | mid + down_blocks.3/.2 |
        # down_blocks.1, leaving only down_blocks.0 (+ the grouped text-state projections at rank 32 + the mapper) for the end of
        # backward: 8.8 MB of 54 MB at rank 32, 30 MB of 543 MB at rank 320.  AQL_LEGS=1 restores round 3's single hook.
        self.n_legs = 3
        self._new_deferred()

    def _new_deferred(self):
        """Descriptor tables of the held-back weight-gradient GEMMs.  Overlapped exchange: one table per backward leg behind a
        router (ops.SplitDeferred) -- region i of the flat gradient buffer, [cuts[i-1], cuts[i]), is complete when backward
        finishes leg i (lora.backward_stage: up path | mid + down_blocks.3/.2 | down_blocks.1) -- and the rest."""
        dev = self.bank.grad.device
        self.deferred = ops.DeferredDW(dev, defer_wide=self.bucketed)
        # the weight-side LoRA form (ops.wside_ok) folds per-sample dY^T X products into dA / dBup / dS between two grouped launches
        # of `flush`: only the single-GPU form of the step flushes in one piece
        self.deferred.allow_post = not (self.split or self.bucketed)
        self.deferred_legs, self.router = [], None
        if self.split:
            cuts = [c for c in self.bank.cuts[:self.n_legs]]
            self.leg_bounds = [0] + cuts                      # leg i owns [leg_bounds[i], leg_bounds[i + 1])
            self.deferred_legs = [ops.DeferredDW(dev) for _ in cuts]
            base = self.bank.grad.data_ptr()
            self.router = ops.SplitDeferred(self.deferred_legs, self.deferred, [base + 4 * b for b in self.leg_bounds])

    @property
    def deferred_early(self):      # the up path's table (leg 0)
        return self.deferred_legs[0] if self.deferred_legs else None

    # ---------------------------------------------------------------------------------------------
    def forward_backward(self, z, msg, eps, t, ctx, flush_dw=True):
        """Everything up to (and including) backward; returns (loss, pred, clean).  With ``flush_dw=False`` the LoRA
        weight-gradient GEMMs stay queued in ``self.deferred`` for `exchange_bucketed`.

        Concurrency: samples are independent (no batch statistics anywhere in the U-Net), so the batch is cut into
        ``micro`` slices that run the whole clean / watermarked / backward chain on their own pair of HIP streams.
        Inside a captured graph these become independent branches that the GPU co-schedules, which fills the CUs
        that a single stream of small-grid kernels leaves idle.  LoRA gradients of all slices accumulate into the
        same flat buffer (fp32 atomics), dS into disjoint rows of one accumulator."""
        B = z.shape[0]
        micro = self.micro if (B % max(self.micro, 1) == 0) else 1
        pro = None
        if self.twin and micro == 1 and _PROLOGUE:
            pro = self._twin_prologue(z, msg, eps, t, ctx)
        if pro is None:
            S = self.mapper(msg)
            if self.ds_accum is None or self.ds_accum.shape != S.shape:
                self.ds_accum = torch.zeros_like(S, dtype=torch.float32)
            self.ds_accum.zero_()
            wm = self.sec_encoder.encode(msg, out_scale=VAE_SCALING)
            x_t, x_t_wm = self.scheduler.add_noise_pair(z, wm, eps, t)
        else:
            S = pro["S"]
        main = torch.cuda.current_stream()
        n = B // micro
        preds, cleans, losses = [], [], []
        ops.DEFERRED = self.router if self.split else self.deferred  # weight-gradient GEMMs + dS reductions are collected ...
        if self.split:
            self._leg_done = [False] * len(self.deferred_legs)
            self.leg_ranges = [[] for _ in self.deferred_legs]
            # fire inside backward, when it finishes leg k (unet.forward registers them on the leg's boundary tensors)
            self.unet._aql_bwd_hooks = [(lambda k=k: self._leg_exchange(k)) for k in range(len(self.deferred_legs))]
        try:
            if self.twin and micro == 1:
                # ONE forward over a twin batch of 2B: clean samples (all-zero scale rows == the reference's clean pass,
                # ppft_train.py:1026-1029) in the first half, watermarked samples in the second; autograd only sees the
                # second half (ops._Dual).  Every kernel of the forward runs once on twice the rows.
                ops.dual_begin()
                try:
                    S_in = S.detach().requires_grad_(True)
                    if pro is not None:   # the prologue kernel wrote the twin buffers; register them and hand out the second halves
                        for k in ("x2", "ctx2", "S16"):
                            ops.DUAL.register(pro[k])
                        x2 = ops.register_cpad(pro["x2"][B:, :z.shape[1]])
                        ctx2 = pro["ctx2"][B:]
                        S_in._aql_s16 = pro["S16"][B:]
                        t_emb = pro["temb"]
                    else:
                        x2 = ops.make_twin(x_t, x_t_wm)
                        c16 = ctx if ctx.dtype == torch.bfloat16 else ctx.to(torch.bfloat16)
                        ctx2 = ops.make_twin(c16, c16)
                        S16 = S.detach().to(torch.bfloat16)
                        S_in._aql_s16 = ops.make_twin(torch.zeros_like(S16), S16)
                        t_emb = None
                    S_in._aql_ds_accum = self.ds_accum
                    pred = self.unet(x2, t, ctx2, cross_attention_kwargs={"scale": S_in}, _aql_t_emb=t_emb).sample
                    clean = ops.clean_twin(pred)
                finally:
                    ops.dual_end()
                loss = ops.mse_loss(pred, clean, unit_grad=True)   # loss.backward() seeds 1.0: d(pred) is used as the kernel wrote it
                loss.backward()
                preds.append(pred.detach())
                cleans.append(clean)
                losses.append(loss.detach())
            else:
                for i in range(micro):
                    sl = slice(i * n, (i + 1) * n)
                    wm_stream, clean_stream = self.streams[2 * i], self.streams[2 * i + 1]
                    wm_stream.wait_stream(main)
                    clean_stream.wait_stream(main)
                    # the frozen "clean" pass is independent of the watermarked pass until the loss
                    with torch.cuda.stream(clean_stream), torch.no_grad():
                        clean = self.unet(x_t[sl], t[sl], ctx[sl], cross_attention_kwargs={"scale": None}).sample
                        ops.join_branches()
                    with torch.cuda.stream(wm_stream):
                        S_in = S[sl].detach().requires_grad_(True)  # the U-Net sees a leaf; all 192 sites accumulate dS
                        S_in._aql_ds_accum = self.ds_accum[sl]       # into one fp32 buffer, pushed through the mapper once
                        pred = self.unet(x_t_wm[sl], t[sl], ctx[sl], cross_attention_kwargs={"scale": S_in}).sample
                        wm_stream.wait_stream(clean_stream)
                        loss = ops.mse_loss(pred, clean)
                        (loss / micro).backward()
                        ops.join_branches()
                    preds.append(pred.detach())
                    cleans.append(clean)
                    losses.append(loss.detach())
                for st in self.streams[:2 * micro]:
                    main.wait_stream(st)
        finally:
            ops.DEFERRED = None
            ops.flush_all_pending()     # (a deferred split-K finalize is always consumed by its GroupNorm: this is the net)
            if self.split:
                self.unet._aql_bwd_hooks = None
        for tns in preds + cleans + losses:
            tns.record_stream(main)
        if self.split:
            self.deferred.flush_ds()
            ops.fold_ds3(self.ds_accum)      # (rank > 32: the grouped q | k | v sites accumulate dS per member, ops.GroupedWideFn)
            S.backward(self.ds_accum)
            self._late_exchange()
        else:
            if flush_dw:
                self.deferred.flush()     # ... and run as two grouped launches here
            else:
                self.deferred.flush_ds()
            ops.fold_ds3(self.ds_accum)
            S.backward(self.ds_accum)
        loss = losses[0] if len(losses) == 1 else torch.stack(losses).mean()   # one twin batch: no copy + mean launches
        if len(preds) == 1:   # one twin batch: hand the tensors out as they are (torch.cat of one tensor is a copy launch each)
            return loss, preds[0], cleans[0]
        return loss, torch.cat(preds), torch.cat(cleans)

    def _twin_prologue(self, z, msg, eps, t, ctx):
        """Everything between the batch and the first GEMM of the twin forward as ONE launch (aql_ppft_prologue): MapperNet, the
        noisy latents of both passes channels-last at conv_in's packed width, the text states twice, the timestep embedding of all 2B
        rows, the [0 | S] bf16 scale rows and the zeroed dS accumulator -- 15 launches of 4-6 us on the generic path (mapper, fill,
        add_noise, three concatenations, the bf16 casts, seven element-wise kernels of the embedding, conv_in's channel padding).
        Returns None when the model / batch is not the SD-1.5 PPFT shape it is written for (the generic path runs)."""
        from .unet import timestep_freq_table
        from .watermark import _MapperGivenFn
        cfg = self.unet.config
        B = z.shape[0]
        if not (z.dim() == 4 and z.shape[1] == 4 and z.dtype == torch.float32 and eps.dtype == torch.float32 and z.is_contiguous()
                and eps.is_contiguous() and t.dtype == torch.int64 and t.numel() == B and ctx.dim() == 3 and ctx.shape[0] == B
                and ctx.is_contiguous() and ctx.dtype in (torch.float32, torch.bfloat16) and (ctx.shape[1] * ctx.shape[2]) % 8 == 0
                and msg.dim() == 2 and self.unet.dtype == torch.bfloat16 and cfg.in_channels == 4):
            return None
        E = self.mapper.bit_embeddings.weight
        bits, r = E.shape
        # raw pointers go to the kernel: everything it indexes must be on this device, dense, and of the width it assumes
        # (acp[t[b]] and E[bit] are read unchecked) -- anything else takes the generic path, which validates / converts
        if not (t.is_cuda and t.is_contiguous() and msg.is_cuda and msg.shape[0] == B and msg.shape[1] == bits and z.is_cuda
                and eps.is_cuda and ctx.is_cuda and eps.shape == z.shape and E.is_cuda and E.is_contiguous()
                and E.dtype == torch.float32):
            return None
        from .lora import _packed_conv3
        if _packed_conv3(self.unet.conv_in).Cin != 8:     # the kernel writes the twin latents at conv_in's packed width of 8
            return None
        dev = z.device
        HW = z.shape[2] * z.shape[3]
        dim = cfg.block_out_channels[0]
        freq = timestep_freq_table(dev, dim)
        wm = self.sec_encoder.encode(msg, out_scale=VAE_SCALING)
        if wm.shape != z.shape:
            return None
        x2 = torch.empty(2 * B, z.shape[2], z.shape[3], 8, dtype=torch.bfloat16, device=dev).permute(0, 3, 1, 2)
        ctx2 = torch.empty((2 * B,) + tuple(ctx.shape[1:]), dtype=torch.bfloat16, device=dev)
        temb = torch.empty(2 * B, dim, dtype=torch.bfloat16, device=dev)
        S32 = torch.empty(B, r, dtype=torch.float32, device=dev)
        S16 = torch.empty(2 * B, r, dtype=torch.bfloat16, device=dev)
        if self.ds_accum is None or self.ds_accum.shape != S32.shape:
            self.ds_accum = torch.zeros_like(S32)
        msg32 = msg.float().contiguous()
        L.call("aql_ppft_prologue", L.ptr(z), L.ptr(wm.float().contiguous()), L.ptr(eps), L.ptr(t), L.ptr(self.scheduler.alphas_cumprod),
               L.ptr(msg32), L.ptr(E), L.ptr(freq), L.ptr(ctx), int(ctx.dtype == torch.float32), B, HW, bits, r, dim // 2,
               ctx.shape[1] * ctx.shape[2], L.ptr(x2), L.ptr(ctx2), L.ptr(temb), L.ptr(S32), L.ptr(S16), L.ptr(self.ds_accum),
               L.stream_ptr())
        S = _MapperGivenFn.apply(msg32, E, S32)
        return {"S": S, "x2": x2, "ctx2": ctx2, "temb": temb, "S16": S16}

    # ------------------------------------------------------------------- overlapped exchange (aql_comm_*, one graph)
    def _tiles(self, dfr, lo, hi):
        """The outputs of the problems queued on ``dfr`` tile [lo, hi) of the flat gradient buffer exactly."""
        base, items = self.bank.grad.data_ptr(), dfr.items
        offs = sorted(((C.data_ptr() - base) // 4, C.numel()) for C, _, _, _ in items)
        pos = lo
        for o, n in offs:
            if o != pos:
                return False
            pos += n
        return pos == hi

    def _leg_exchange(self, k):
        """Called from the U-Net's backward hook k (unet._aql_bwd_hooks): backward has finished leg k, every LoRA site of that
        leg has queued its weight-gradient GEMMs, whose outputs are region [leg_bounds[k], leg_bounds[k+1]) of the flat gradient
        buffer.  Launch them now and fork the side stream behind them: the all-reduce(mean) of that region runs under the rest
        of backward (DDP's grad-ready buckets, ppft_train.py:1058).  Inside a capture this becomes a branch of the graph."""
        if self._leg_done[k]:
            return
        for j in range(k):             # hooks fire in leg order; a leg whose boundary tensor carried no gradient is flushed here
            self._leg_exchange(j)
        self._leg_done[k] = True
        b, e = self.bank, self.deferred_legs[k]
        lo0, hi0 = self.leg_bounds[k], self.leg_bounds[k + 1]
        if not e.items:
            return
        if not self._tiles(e, lo0, hi0):
            raise L.AqlError(f"overlapped exchange: the weight gradients of backward leg {k} do not tile their region of the gradient buffer")
        ranges = e.plan(b.grad, dp.bucket_count(4 * (hi0 - lo0)) if self.overlap else 1)
        main = torch.cuda.current_stream()
        for q, (lo, hi) in enumerate(ranges):
            e.run_bucket(q)                       # on the backward stream
            if self.overlap:
                self.side.wait_stream(main)       # fork: the collective of this bucket runs under what follows on `main`
                with torch.cuda.stream(self.side):
                    self.comm.all_reduce_(b.grad[lo:hi], average=True)
        self.leg_ranges[k] = ranges

    def _early_exchange(self):
        self._leg_exchange(0)

    @property
    def early_ranges(self):        # every bucket issued from a backward hook, in issue order
        return [r for leg in self.leg_ranges for r in leg]

    def _late_exchange(self):
        """End of backward: the remaining weight gradients in buckets on the main stream, each bucket's all-reduce on the
        side stream under the next bucket's GEMMs; the mapper gradient (complete after S.backward) rides with the last one."""
        b, d = self.bank, self.deferred
        main = torch.cuda.current_stream()
        for k in range(len(self.deferred_legs)):      # a hook that never fired (cannot happen on the PPFT path): do it now
            self._leg_exchange(k)
        lo0 = 0
        for k, e in enumerate(self.deferred_legs):    # the late region starts behind the last leg that queued anything
            if e.items:
                lo0 = self.leg_bounds[k + 1]
        if not self._tiles(d, lo0, b.n_lora):
            raise L.AqlError("overlapped exchange: the weight gradients do not tile the gradient buffer")
        ranges = d.plan(b.grad, dp.bucket_count(4 * (b.n_lora - lo0)) if self.overlap else 1)
        ranges[-1] = (ranges[-1][0], b.numel)          # + the mapper gradient
        for k, (lo, hi) in enumerate(ranges):
            d.run_bucket(k)
            if self.overlap:
                self.side.wait_stream(main)
                with torch.cuda.stream(self.side):
                    self.comm.all_reduce_(b.grad[lo:hi], average=True)
        main.wait_stream(self.side)
        self.late_ranges = ranges
        for e in self.deferred_legs:
            e.reset()
        d.reset()

    def exchange_gradients(self):
        """DDP's gradient all-reduce(mean) (accelerator.backward, ppft_train.py:1058) as ONE collective over the flat
        fp32 gradient buffer (54 MB at r=32, 543 MB at r=320).  Used when the weight gradients were already flushed."""
        dp.allreduce_mean_(self.bank.grad[:self.bank.numel], self.pg)

    def plan_exchange(self):
        """Bucket plan for the queued weight-gradient GEMMs: [(lo, hi)] ranges tiling [0, n_lora) of the flat gradient
        buffer.  The bank is laid out in gradient-ready order, so bucket 0 holds the sites backward reached first."""
        b = self.bank
        ranges = self.deferred.plan(b.grad, dp.bucket_count(4 * b.n_lora))
        if ranges:
            ranges[0] = (0, ranges[0][1])
            ranges[-1] = (ranges[-1][0], b.n_lora)
        return ranges

    def exchange_bucketed(self, ranges, run_bucket):
        """Overlapped exchange (DDP's grad-ready buckets, ppft_train.py:1058): the mapper gradient is complete after
        backward and goes first; then for every bucket k: launch its weight-gradient GEMMs (``run_bucket(k)``) and hand
        grad[lo_k:hi_k] to RCCL, which runs on the collective stream under bucket k+1's GEMMs."""
        b = self.bank
        self.reducer.launch(b.grad[b.n_lora:b.numel])
        for k, (lo, hi) in enumerate(ranges):
            run_bucket(k)
            self.reducer.launch(b.grad[lo:hi])
        self.reducer.finish()

    def optimizer_step(self):
        b = self.bank
        b1, b2, eps, wd = self.hp
        n_lora, n_all = b.n_lora, b.numel
        self.step_t += 1
        st = L.stream_ptr()
        L.call("aql_sumsq_f32", L.ptr(b.grad), n_lora, L.ptr(self.sumsq), st)
        L.call("aql_clipnorm_adamw", L.ptr(b.flat), L.ptr(b.grad), L.ptr(b.exp_avg), L.ptr(b.exp_avg_sq), n_lora,
               L.ptr(self.sumsq), float(self.max_grad_norm), L.ptr(self.lr_t), b1, b2, eps, wd, L.ptr(self.step_t), st)
        off = n_lora
        L.call("aql_clipnorm_adamw", L.ptr(b.flat[off:]), L.ptr(b.grad[off:]), L.ptr(b.exp_avg[off:]),
               L.ptr(b.exp_avg_sq[off:]), n_all - n_lora, None, 0.0, L.ptr(self.lr_t), b1, b2, eps, wd,
               L.ptr(self.step_t), st)
        b.refresh()
        b.zero_grad()

    def _step_body(self, z, msg, eps, t, ctx):
        if self.split:
            loss, _, _ = self.forward_backward(z, msg, eps, t, ctx)    # the exchange is part of it
        elif self.bucketed:
            loss, _, _ = self.forward_backward(z, msg, eps, t, ctx, flush_dw=False)
            self.exchange_bucketed(self.plan_exchange(), self.deferred.run_bucket)
            self.deferred.reset()
        else:
            loss, _, _ = self.forward_backward(z, msg, eps, t, ctx)
            self.exchange_gradients()
        self.optimizer_step()
        return loss

    def step(self, z, msg, eps, t, ctx):
        loss = self._step_body(z, msg, eps, t, ctx)
        self.global_step += 1
        self.lr_t.fill_(self.base_lr * self.lr_lambda(self.global_step))
        return loss

    # ---------------------------------------------------------------------------------------------
    def capture(self, batch, warmup=2):
        """Capture the step into HIP graphs (torch.cuda.CUDAGraph == hipGraph on ROCm): one graph for
        forward+backward, one for clip+AdamW+weight re-cast; the gradient all-reduce runs between them on the same
        stream.  ~3.7k kernel launches per step become two graph launches, which removes the host from the critical
        path.  Returns ``run(z, msg, eps, t, ctx) -> loss`` that copies the inputs into static buffers and replays."""
        static = {k: v.clone() for k, v in batch.items()}
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._step_body(**static)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g_fb, g_opt = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        if self.bucketed:
            return self._capture_bucketed(static, g_fb, g_opt)
        import os
        if self.split or not dp.exchange_active(self.pg):
            return self._capture_single(static, g_fb)
        # thread_local capture mode: with a process group alive, RCCL's watchdog thread polls hipEventQuery on the
        # warm-up collectives; under the default global mode that call is illegal while ANY thread captures and aborts
        # the process ("operation not permitted when stream is capturing") -- found with AQL_FORCE_ALLREDUCE=1.
        # the captured memcpy nodes re-read the pinned descriptor tables at every replay: the capture gets its own
        # DeferredDW so that a later eager step() cannot rewrite them
        eager_deferred = self.deferred
        self.deferred = ops.DeferredDW(eager_deferred.device, defer_wide=False)
        self.deferred.allow_post = eager_deferred.allow_post
        with torch.cuda.graph(g_fb, capture_error_mode="thread_local"):
            loss, _, _ = self.forward_backward(**static)
        with torch.cuda.graph(g_opt, pool=g_fb.pool(), capture_error_mode="thread_local"):
            self.optimizer_step()
        captured = self.deferred
        self.deferred = eager_deferred
        self._graphs = (g_fb, g_opt, static, loss, captured)

        def run(z, msg, eps, t, ctx):
            _feed(static, z, msg, eps, t, ctx)
            g_fb.replay()
            self.exchange_gradients()
            g_opt.replay()
            self.global_step += 1
            self.lr_t.fill_(self.base_lr * self.lr_lambda(self.global_step))
            return loss

        run.is_graph = True
        return run

    def _capture_single(self, static, g):
        """The whole step as ONE HIP graph: forward + backward (+ the gradient exchange as a forked branch: the early buckets
        under the mid / down backward, the rest behind the last weight-gradient launch, all through aql_comm_* on the side
        stream) + clip + AdamW + re-cast.  Single GPU: the same graph without collectives."""
        eager = (self.deferred, self.deferred_legs, self.router)
        self._new_deferred()          # the captured memcpy nodes re-read these pinned tables at every replay
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            loss, _, _ = self.forward_backward(**static)
            self.optimizer_step()
        captured = (self.deferred, self.deferred_legs, self.router)
        self.deferred, self.deferred_legs, self.router = eager
        self._graphs = (g, static, loss, captured)

        def run(z, msg, eps, t, ctx):
            _feed(static, z, msg, eps, t, ctx)
            g.replay()
            self.global_step += 1
            self.lr_t.fill_(self.base_lr * self.lr_lambda(self.global_step))
            return loss

        run.is_graph = True
        run.n_graphs = 1
        return run

    def _capture_bucketed(self, static, g_fb, g_opt):
        """Data-parallel capture: graph 0 = forward + backward (+ dS, mapper backward); one small graph per gradient
        bucket (its grouped / wide weight-gradient GEMMs); graph N+1 = clip + AdamW + re-cast.  The RCCL collectives
        are issued eagerly between the bucket graphs (RCCL calls are kept out of the captures on purpose: a captured
        collective cannot be validated on the 1-GPU boxes this is developed on)."""
        eager_deferred = self.deferred
        self.deferred = ops.DeferredDW(eager_deferred.device, defer_wide=True)  # its pinned table belongs to the graphs
        with torch.cuda.graph(g_fb, capture_error_mode="thread_local"):
            loss, _, _ = self.forward_backward(**static, flush_dw=False)
        ranges = self.plan_exchange()
        g_dw = []
        for k in range(len(ranges)):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=g_fb.pool(), capture_error_mode="thread_local"):
                self.deferred.run_bucket(k)
            g_dw.append(g)
        with torch.cuda.graph(g_opt, pool=g_fb.pool(), capture_error_mode="thread_local"):
            self.optimizer_step()
        captured = self.deferred      # keeps the operands of the bucket graphs (capture-pool memory) referenced
        self.deferred = eager_deferred
        self._graphs = (g_fb, g_dw, g_opt, static, loss, captured)
        self.exchange_ranges = ranges

        import os
        run_bucket = lambda k: g_dw[k].replay()   # noqa: E731

        def run(z, msg, eps, t, ctx):
            _feed(static, z, msg, eps, t, ctx)
            g_fb.replay()
            self.exchange_bucketed(ranges, run_bucket)
            g_opt.replay()
            self.global_step += 1
            self.lr_t.fill_(self.base_lr * self.lr_lambda(self.global_step))
            return loss

        run.is_graph = True
        return run

    def grad_norm(self):
        return float(torch.sqrt(self.sumsq)[0])

    def logged_loss(self, loss):
        """The cross-rank mean the reference logs (ppft_train.py:1054); synchronises, call it at logging steps only."""
        return dp.logged_loss(loss, self.pg)
