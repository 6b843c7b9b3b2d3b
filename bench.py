"""PPFT train-step benchmark (BASELINE.json metric): images/sec at 512x512 (64x64x4 latents), SD-1.5 U-Net,
watermark-LoRA rank 32, 48-bit messages, batch 4 per GPU, bf16 -- synthetic latents/text states, random-init weights.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

One JSON line on rank 0.  `roofline` prices the whole captured step against the dense bf16 MFMA peak with the
algorithmic FLOPs of SURVEY.md §8(d) (2.477 TFLOP/image at r=32); `kernels` adds HIP-event timings of the two
dominant kernels in isolation.  `cpu_baseline` times the CPU oracle on this box's host cores (baseline only).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TF = 2500.0  # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md
UNET_FWD_GFLOP = 803.3
LORA_FWD_GFLOP_PER_RANK = 0.7006


def step_tflop_per_image(rank):
    return (3 * UNET_FWD_GFLOP + 3 * LORA_FWD_GFLOP_PER_RANK * rank) / 1e3


def build(device, rank, seed=2048, micro=1):
    from aqualora_amd import synth
    from aqualora_amd.lora import inject_lora
    from aqualora_amd.ppft import PPFTTrainer
    from aqualora_amd.unet import UNet2DConditionModel, init_synthetic, lora_keys
    from aqualora_amd.watermark import MapperNet, SecretEncoder, cosine_lr_lambda
    unet = UNet2DConditionModel(device=device, dtype=torch.bfloat16)
    init_synthetic(unet, seed)
    keys = lora_keys(unet)
    inject_lora(unet, rank, keys)
    with torch.no_grad():
        for k in keys:
            lay = unet.get_submodule(k).lora_layer
            lay.down.weight.copy_(synth.normal(k + ".lora.down", lay.down.weight.shape, 1.0 / rank, seed, device))
            lay.up.weight.copy_(synth.normal(k + ".lora.up", lay.up.weight.shape, 0.02, seed, device))
    mapper = MapperNet(48, rank)
    enc = SecretEncoder(48)
    with torch.no_grad():
        enc.secret_scaler[5].weight.copy_(synth.normal("enc.conv.w", (4, 4, 3, 3), 0.05, seed))
    tr = PPFTTrainer(unet, mapper, enc, rank, learning_rate=1e-4,
                     lr_lambda=cosine_lr_lambda(0, 100000, lr_end=0.01), micro_batches=micro)
    return tr


def synthetic_batch(B, device, rank_id, seed=2048, step=0):
    """One synthetic batch, resident in HBM.  ``step`` selects an independent draw (fresh latents, messages, noise and
    timesteps for every step, like a data loader would deliver them)."""
    from aqualora_amd import synth
    s = seed + 977 * rank_id
    tag = "" if step == 0 else f".{step}"
    return dict(z=synth.normal("bench.z" + tag, (B, 4, 64, 64), 1.0, s, device),
                msg=synth.bits("bench.msg" + tag, (B, 48), s, device),
                eps=synth.normal("bench.eps" + tag, (B, 4, 64, 64), 1.0, s, device),
                t=synth.randint("bench.t" + tag, (B,), 1000, s, device),
                ctx=synth.normal("bench.ctx" + tag, (B, 77, 768), 1.0, s, device).to(torch.bfloat16))


def time_kernel(fn, iters=20):
    """GPU time of one call of fn -> (ms, how).  `iters` calls are captured into one HIP graph and the replay is bracketed by HIP
    events on torch's current stream (the stream the C-ABI launches go to) -- eager calls of a ~50 us kernel are bound by the
    Python / ctypes call overhead, not by the kernel.  Reported: the MINIMUM over 3 replays of the per-call average (`how` says
    so, and says "eager" if the capture was refused and the calls were timed one by one instead)."""
    fn()
    torch.cuda.synchronize()
    how = f"hip-graph replay of {iters} launches, min of 3 replays"
    gr = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(gr):
            for _ in range(iters):
                fn()
        run = gr.replay
    except RuntimeError as e:   # capture refused: torch.cuda.graph's __exit__ has ended the capture; time the eager calls
        torch.cuda.synchronize()
        how = f"EAGER ({iters} launches, min of 3; graph capture failed: {str(e)[:80]})"

        def run():
            for _ in range(iters):
                fn()
    run()
    torch.cuda.synchronize()
    best = None
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        best = ms if best is None else min(best, ms)
    return best, how


def dominant_kernels(B, device):
    """HIP-event timings (torch's current stream == the launch stream of the C-ABI calls) of the heaviest launch of each
    kernel family, at the shapes the step runs them.  kernels[0] is the heaviest launch of the TIME-dominant family (the
    one-launch LoRA linears: 34 % of the step, profiles/r06_families_config2.json): ff.net.0.proj + rank-32 LoRA + GEGLU on the twin
    batch of the 64x64 level.  The forward runs both U-Net passes as ONE twin batch of 2B samples (ops._Dual), so forward
    launches see 2B samples; rows of the clean half skip the LoRA branch (lora_row0)."""
    from aqualora_amd import _lib as L
    from aqualora_amd import ops, synth
    out = []

    def entry(kernel, ms_how, fl, **kw):
        ms, how = ms_how
        d = {"kernel": kernel, "ms": ms, "timing": how, "flops": fl, "achieved_tflops": fl / ms / 1e9,
             "frac_of_mfma_peak": fl / ms / 1e9 / MFMA_PEAK_TF}
        d.update(kw)
        out.append(d)

    # (0) ff.net.0.proj 320 -> 2x1280 + LoRA + GEGLU, twin batch: M = 2B x 4096 rows, LoRA on rows >= M/2, the pre-activation H
    # written for those rows only (what FeedForwardFn's backward needs)
    M, K, F = 2 * B * 4096, 320, 1280
    X = synth.normal("k.gx", (M, K), 1.0, 1, device).to(torch.bfloat16)
    W = synth.normal("k.gw", (2 * F, K), K ** -0.5, 1, device).to(torch.bfloat16)
    bias = synth.normal("k.gb", (2 * F,), 0.02, 1, device).to(torch.bfloat16)
    A = synth.normal("k.ga", (32, K), 1.0 / 32, 1, device).to(torch.bfloat16)
    Bu = synth.normal("k.gu", (2 * F, 32), 0.02, 1, device).to(torch.bfloat16)
    S = torch.cat([torch.zeros(B, 32, device=device), synth.normal("k.gs", (B, 32), 1.0, 1, device)]).to(torch.bfloat16)
    H = torch.empty(M, 2 * F, dtype=torch.bfloat16, device=device)
    G = torch.empty(M, F, dtype=torch.bfloat16, device=device)
    T = torch.empty(M, 32, dtype=torch.bfloat16, device=device)
    Ts = torch.empty_like(T)

    def geglu_call():
        rc = L.call_raw("aql_lora_gemm_fused_geglu", L.ptr(X), K, L.ptr(W), K, M, F, K, L.ptr(A), L.ptr(S), 4096, L.ptr(Bu),
                        L.ptr(bias), L.ptr(H), 2 * F, L.ptr(G), F, L.ptr(T), L.ptr(Ts), M // 2, L.stream_ptr())
        assert rc == 0, rc
    fl = 2.0 * M * K * 2 * F + (M // 2) * 2.0 * (K * 32 + 32 * 2 * F)
    alg = 2.0 * (M * K + 2 * F * K + M * F + (M // 2) * 2 * F + 2 * (M // 2) * 32)
    entry("lora_geglu256_kernel (256x256 persistent tile, GEGLU epilogue) ff.net.0.proj 320->2x1280 + rank-32 LoRA + GEGLU, "
          f"{2 * B} samples x 4096 tokens (twin forward)", time_kernel(geglu_call), fl, samples=2 * B, algorithmic_bytes=alg,
          pmc_key=f"lora_geglu 320->2x1280 M={M}")
    del X, W, H, G, T, Ts
    # (1, 2) the FLOP-dominant family: 3x3 conv 320->320 @64x64 (48 % of the step's algorithmic FLOPs)
    w = synth.normal("k.w", (320, 320, 3, 3), 0.02, 1, device)
    pk = ops.PackedConv3x3(w, torch.zeros(320, device=device), 1)
    for nb, tag in ((2 * B, "twin forward"), (B, "batch-B shape")):
        x = synth.normal("k.x", (nb, 320, 64, 64), 1.0, 1, device).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        with torch.no_grad():
            th = time_kernel(lambda: ops.conv3x3(x, pk))
        fl = 2.0 * nb * 64 * 64 * 320 * 9 * 320
        # 2B samples: 256 tiles of 256x160 on the 12-wave kernel (one chip-wide round); B samples: 256 tiles of 128x160, 8 waves
        # (also two whole rounds of 256x160 tiles: config 3's twin forward; the picker is aql_gemm.hip launch_cfg)
        t256 = nb * 4096 // 256 * 2
        kname = ("conv_row_kernel (256x160 row tile, 12 waves)" if t256 in range(240, 257) or t256 == 512
                 else "conv_row_kernel (128x160 row tile, 8 waves)")
        entry(f"{kname} conv3x3 320->320 @64x64, {nb} samples ({tag})", th, fl, samples=nb,
              algorithmic_bytes=2.0 * nb * 64 * 64 * 320 * 2 + 320 * 2880 * 2, pmc_key=f"conv3x3 320->320 @64x64 B={nb}")
    # (3) the "attention GEMMs" of the north star: flash attention at the 64x64 level (8 heads of 40, 4096 tokens) ...
    qkv = [synth.normal(f"k.{n}", (B, 4096, 320), 1.0, 1, device).to(torch.bfloat16) for n in "qkv"]
    with torch.no_grad():
        th = time_kernel(lambda: ops.attention(qkv[0], qkv[1], qkv[2], 8))
    entry("attn_fwd_kernel self-attention 8 heads x 40, 4096 tokens", th, 4.0 * B * 8 * 4096 * 4096 * 40, samples=B)
    # (4) ... and the one-launch rank-32 LoRA linear of the same level (to_out / attn2.to_q / proj_in / proj_out, 320 -> 320)
    X = synth.normal("k.lx", (B * 4096, 320), 1.0, 1, device).to(torch.bfloat16)
    Wl = synth.normal("k.lw", (320, 320), 0.05, 1, device).to(torch.bfloat16)
    Al = synth.normal("k.la", (32, 320), 0.05, 1, device).to(torch.bfloat16)
    Bl = synth.normal("k.lb", (320, 32), 0.05, 1, device).to(torch.bfloat16)
    Sl = synth.normal("k.ls", (B, 32), 1.0, 1, device).to(torch.bfloat16)
    Yl = torch.empty(B * 4096, 320, dtype=torch.bfloat16, device=device)
    Tl = torch.empty(B * 4096, 32, dtype=torch.bfloat16, device=device)
    Tsl = torch.empty_like(Tl)

    def lora_call():
        rc = L.call_raw("aql_lora_gemm_fused", L.ptr(X), 320, L.ptr(Wl), 320, B * 4096, 320, 320, L.ptr(Al), L.ptr(Sl), 4096,
                        L.ptr(Bl), None, None, 0, L.ptr(Yl), 320, L.ptr(Tl), L.ptr(Tsl), 0, L.stream_ptr())
        assert rc == 0, rc
    Ml = B * 4096
    entry("lora_gemm_kernel attention projection 320->320 + rank-32 LoRA @4096 tok (backward-data shape)", time_kernel(lora_call),
          2.0 * Ml * 320 * (320 + 32) + 2.0 * Ml * 32 * 320, samples=B,
          algorithmic_bytes=2.0 * (2 * Ml * 320 + 2 * Ml * 32 + 320 * 320 + 2 * 32 * 320))
    # (5) the attention projections of the 64 x 64 level in the FORWARD pass: the row-resident chain  attn1.to_out + residual -> norm2 ->
    # attn2.to_q  (two LoRA linears + LayerNorm, one launch, twin batch; csrc/aql_chain.hip).  168 FLOP per byte: HBM-bound by shape --
    # its roof is the 8 TB/s of `hbm_view`, not the matrix pipe
    Mc, C = 2 * B * 4096, 320
    rb = lambda n, *sh: synth.normal(n, sh, 0.05, 1, device).to(torch.bfloat16)   # noqa: E731
    eb = lambda *sh: torch.empty(*sh, dtype=torch.bfloat16, device=device)       # noqa: E731
    Xc, Rc = synth.normal("k.cx", (Mc, C), 1.0, 1, device).to(torch.bfloat16), synth.normal("k.cr", (Mc, C), 1.0, 1, device).to(torch.bfloat16)
    Sc = torch.cat([torch.zeros(B, 32, device=device), synth.normal("k.cs", (B, 32), 1.0, 1, device)]).to(torch.bfloat16)
    lin = lambda t, bias: dict(W=rb(t + "w", C, C), bias=rb(t + "b", C) if bias else None, Ad=rb(t + "a", 32, C), Bup=rb(t + "u", C, 32), ldw=C)   # noqa: E731
    stg = [dict(lin("k.c0", True), T=eb(Mc, 32), Ts=eb(Mc, 32), res=Rc, ldr=C, out=eb(Mc, C), ldo=C, keep=1, ln=1, gamma=1 + rb("k.cg", C),
                beta=rb("k.cb", C), eps=1e-5, stats=torch.empty(Mc, 2, device=device), nout=eb(Mc, C), ldn=C, nout_row0=Mc // 2),
           dict(lin("k.c1", False), T=eb(Mc, 32), Ts=eb(Mc, 32), out=eb(Mc, C), ldo=C, keep=0)]
    th = time_kernel(lambda: ops.chain_fwd(Xc, C, Mc, 4096, Mc // 2, Sc, stg))
    fl = 2 * (2.0 * Mc * C * C) + 2 * (Mc // 2) * 2.0 * (C * 32 + 32 * C)
    # x + residual in, hs + q out on all rows, the LayerNorm output and T / Ts of both linears on the watermarked half, weights
    alg = 2.0 * (2 * Mc * C + 2 * Mc * C + (Mc // 2) * C + 4 * (Mc // 2) * 32 + 2 * C * C + 4 * 32 * C) + 8.0 * Mc
    entry(f"chain_kernel<2,false> attn1.to_out + residual -> norm2 -> attn2.to_q (row-resident chain, rank-32 LoRA on both linears), "
          f"{2 * B} samples x 4096 tokens (twin forward)", th, fl, samples=2 * B, algorithmic_bytes=alg,
          pmc_key=f"chain a: to_out+res -> LN -> to_q, M={Mc} (twin)")
    for d in out:   # every launch also against the HBM roof (algorithmic bytes / measured time)
        if "algorithmic_bytes" in d:
            d["hbm_view"] = {"achieved_GBps": d["algorithmic_bytes"] / d["ms"] / 1e6, "peak_GBps": 8000.0,
                             "frac": d["algorithmic_bytes"] / d["ms"] / 1e6 / 8000.0}
    return out


def load_profile_json(name):
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None
    with open(path) as f:
        return json.load(f)


def profile_staleness():
    """Which kernel sources changed since the committed profiles (families, PMC traffic, MfmaUtil) were taken: those sections of the
    line are read from files, not measured in this run, and describe an older library when this list is not empty."""
    import hashlib
    meta = load_profile_json("r06_meta.json")
    if meta is None:
        return {"profile_meta": "missing"}
    changed = []
    for f, h in meta["files"].items():
        path = os.path.join(ROOT, "aqualora_amd", "csrc", f)
        cur = hashlib.sha256(open(path, "rb").read()).hexdigest()[:16] if os.path.exists(path) else None
        if cur != h:
            changed.append(f)
    return {"kernel_sources_changed_since_profile": changed, "static_sections_current": not changed}


def exchange_record(tr, device, world, iters=10):
    """The gradient exchange ALONE, per rank (HIP events on the launch stream around the collective; the payload is the flat fp32
    gradient buffer of the trainer): what the step's exchange costs when nothing hides it, and the bus bandwidth it implies for a ring
    all-reduce, 2 (N - 1) / N x bytes / time.  Both forms are timed when both exist: the bucketed torch.distributed all-reduce (the
    default) and aql_comm_all_reduce_f32 on the caller's stream (AQL_COMM=1).  With ONE rank RCCL does no transfer: the numbers
    are the collective's launch floor, not a bandwidth."""
    from aqualora_amd import dp
    flat = tr.bank.grad[:tr.bank.numel]
    nbytes = flat.numel() * 4
    keep = flat.clone()
    out = {"payload_bytes": nbytes, "ranks": world, "iters": iters}

    def time_it(fn):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        per_rank = [None] * world
        if world > 1:
            dist.all_gather_object(per_rank, ms)
        else:
            per_rank = [ms]
        worst = max(per_rank)
        return {"exchange_ms_per_rank": per_rank, "exchange_ms": worst,
                "bus_GBps": (2.0 * (world - 1) / world * nbytes / (worst * 1e-3) / 1e9) if world > 1 else None}

    if dp.exchange_active(tr.pg):
        out["torch_distributed_allreduce"] = time_it(lambda: dp.allreduce_mean_(flat, tr.pg))
        comm = tr.comm
        if comm is not None:
            out["aql_comm_allreduce"] = time_it(lambda: comm.all_reduce_(flat, average=True))
        else:
            out["aql_comm_allreduce"] = None
            out["aql_comm_note"] = tr.comm_note
    else:
        out["note"] = "no exchange active (single process without AQL_FORCE_ALLREDUCE)"
    flat.copy_(keep)
    return out


def config3_record(device, rank_id, steps=10, warmup=3):
    """BASELINE config 3 per GPU (rank 320, batch 8; train/README.md:34-48) on this GPU, as a sub-record of the default line:
    the same captured step, HIP-event median over `steps` replays with a fresh batch each."""
    rank, B = 320, 8
    tr = build(device, rank)
    batches = [synthetic_batch(B, device, rank_id, step=i) for i in range(4)]
    run = tr.capture(batches[0])
    for i in range(warmup):
        run(**batches[i % 4])
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for i in range(steps):
        evs[i][0].record()
        loss = run(**batches[i % 4])
        evs[i][1].record()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    per = sorted(a.elapsed_time(b) for a, b in evs)
    tf = step_tflop_per_image(rank) * B
    rec = {"workload": f"SD1.5 PPFT LoRA rank={rank}, batch={B}/GPU (BASELINE config 3 per GPU), latent-in, 1 GPU",
           "value": B / dt, "unit": "images/sec", "ms_per_step": 1e3 * dt, "ms_per_step_hip_event_median": per[len(per) // 2],
           "steps": steps, "warmup": warmup, "loss": float(loss), "hip_graph": bool(getattr(run, "is_graph", False)),
           "step_roofline": {"bound": "mfma", "achieved": tf / dt, "peak": MFMA_PEAK_TF, "unit": "TFLOP/s",
                             "frac": tf / dt / MFMA_PEAK_TF,
                             "launch": f"one PPFT step = {step_tflop_per_image(rank):.3f} TFLOP/image x {B} images"}}
    del tr, run
    torch.cuda.empty_cache()
    return rec


def cpu_baseline(tr, rank):
    """Bounded CPU sample of the SAME workload: ONE real PPFT step of the oracle (oracle/ppft_oracle.py: clean forward,
    watermarked forward with the LoRA branch, MSE, backward to all 384 LoRA tensors + the mapper) on the full-size
    SD-1.5 U-Net at batch 1, fp32, on the host cores -- about 10-20 s after one warm-up forward."""
    from oracle import ppft_oracle as O
    from aqualora_amd.unet import SD15, lora_keys
    from aqualora_amd import synth
    cores = min(32, len(os.sched_getaffinity(0)))  # more threads than this slows torch's CPU convs down on big hosts
    torch.set_num_threads(cores)
    sd = {k: v.detach().float().cpu() for k, v in tr.unet.state_dict().items() if "lora_layer" not in k}
    keys = lora_keys(tr.unet)
    lora = {}
    for k in keys:
        layer = tr.unet.get_submodule(k).lora_layer
        lora[k] = (layer.down.weight.detach().float().cpu().clone().requires_grad_(True),
                   layer.up.weight.detach().float().cpu().clone().requires_grad_(True))
    E = tr.mapper.bit_embeddings.weight.detach().float().cpu().clone().requires_grad_(True)
    z = synth.normal("bench.z", (1, 4, 64, 64), 1.0, 2048)
    wm = synth.normal("bench.wm", (1, 4, 64, 64), 0.05, 2048)
    eps = synth.normal("bench.eps", (1, 4, 64, 64), 1.0, 2048)
    msg = synth.bits("bench.msg", (1, 48), 2048)
    ctx = synth.normal("bench.ctx", (1, 77, 768), 1.0, 2048)
    t = torch.tensor([500])
    with torch.no_grad():  # warm-up: pages the weights in, spins the thread pool up
        t0 = time.perf_counter()
        O.UNetOracle(sd, dict(SD15)).forward(z, t, ctx, None)
        fwd_s = time.perf_counter() - t0
    times = []
    for _ in range(3):   # 1 warm-up + 3 timed steps, median
        for pr in lora.values():
            for p in pr:
                p.grad = None
        E.grad = None
        t0 = time.perf_counter()
        loss, _, _, _ = O.ppft_loss(sd, dict(SD15), lora, E, msg, z, wm, eps, t, ctx)
        loss.backward()
        times.append(time.perf_counter() - t0)
    step_s = sorted(times)[1]
    return {"value": 1.0 / step_s, "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"median of 3 full PPFT steps of oracle/ppft_oracle.py (clean fwd + LoRA fwd + backward to {2 * len(keys)} "
                      f"LoRA tensors) on the full-size SD-1.5 U-Net, batch 1, fp32, {cores} threads: "
                      + " / ".join(f"{x:.2f}" for x in times) + f" s (after a warm-up forward of {fwd_s:.2f} s)"}


def vae_bench(args, device):
    """Frozen VAE encode (ppft_train.py:993) and decode at 512x512 on the HIP kernels: SURVEY.md §8 (f) rank 1.  Not part
    of the headline step (latent-in); reported so that a pixel-in step can be priced: step + batch / encode rate."""
    from aqualora_amd import synth
    from aqualora_amd.vae import SD15_VAE, AutoencoderKL, encode_gflop, synthetic_state_dict
    vae = AutoencoderKL(synthetic_state_dict(SD15_VAE, device=device), SD15_VAE, device)
    B = args.batch
    x = synth.normal("vae.x", (B, 3, 512, 512), 0.5, 1, device).clamp(-1, 1)
    noise = synth.normal("vae.n", (B, 4, 64, 64), 1.0, 1, device)
    z = vae.encode(x, noise)
    vae.decode(z)
    torch.cuda.synchronize()
    n = max(2, args.steps // 2)
    t0 = time.perf_counter()
    for _ in range(n):
        z = vae.encode(x, noise)
    torch.cuda.synchronize()
    te = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(n):
        img = vae.decode(z)
    torch.cuda.synchronize()
    td = (time.perf_counter() - t0) / n
    gf = encode_gflop()
    print(json.dumps({"metric": "frozen SD-1.5 VAE encode images/sec at 512x512", "value": B / te, "unit": "images/sec",
                      "n_gpus": 1, "steps": n, "ms_per_batch": 1e3 * te, "dtype": "bf16", "higher_is_better": True,
                      "data": "synthetic", "config": {"workload": f"AutoencoderKL.encode, batch {B}, 3x512x512 -> 4x64x64"},
                      "roofline": {"bound": "mfma", "achieved": gf * B / te / 1e3, "peak": MFMA_PEAK_TF, "unit": "TFLOP/s",
                                   "frac": gf * B / te / 1e3 / MFMA_PEAK_TF, "gflop_per_image": gf},
                      "decode": {"images_per_sec": B / td, "ms_per_batch": 1e3 * td,
                                 "finite": bool(torch.isfinite(img).all())}}), flush=True)


def synthetic_decoder(bits, device, seed=2048, tag="rob."):
    """SecretDecoder (EfficientNet-B1 + Linear(1280, 2*bits), utils/models.py:84-96) with synthetic weights: He-normal convolutions,
    BatchNorm gamma 1 / beta 0 / running statistics (0, 1).  There is no ImageNet checkpoint on the box."""
    from aqualora_amd import synth
    from aqualora_amd.decoder import SecretDecoder
    dec = SecretDecoder(bits)
    with torch.no_grad():
        for name, t in list(dec.named_parameters()) + list(dec.named_buffers()):
            if name.endswith("running_var"):
                t.fill_(1.0)
            elif name.endswith("running_mean") or name.endswith("num_batches_tracked"):
                t.zero_()
            elif name.endswith(".1.weight") and t.dim() == 1:
                t.fill_(1.0)
            elif t.dim() == 1:
                t.zero_()
            else:
                fan = t[0].numel()
                t.copy_(synth.normal(tag + name, tuple(t.shape), (2.0 / fan) ** 0.5, seed))
    return dec.to(device)


def oracle_bits(dec, images):
    """The CHECKER: message bits of `images` from the CPU restatement of SecretDecoder.forward (oracle/decoder_oracle.py) on the
    decoder's own state dict.  Returns (bits [n, k] int64, logits [n, k, 2])."""
    from oracle import decoder_oracle as DO
    sd = {k: v.detach().float().cpu() for k, v in dec.state_dict().items()}
    with torch.no_grad():
        logits = DO.secret_decoder(sd, images.detach().float().cpu(), dec.output_size)
    return logits.argmax(-1), logits


def bit_accuracy_record(device, n_images=4, seed=2048):
    """The second half of BASELINE.json's metric on the extraction path (evaluation/utils_eval.py:131-140,156-213): the HIP
    SecretDecoder's 48 bits per 512x512 image against the CPU oracle's bits on the SAME images and weights (fraction equal: must be
    1.0 -- "extracted message bits bit-exact"), the bit accuracy / TPR arithmetic of utils_eval.py:199-213 on them, and the
    extraction rate at batch 1 (the reference decodes image by image) and batch 16.  Weights are synthetic, so the accuracy against
    a ground-truth message is the agreement with the oracle's decode, not a trained watermark's (that is tests/test_roundtrip.py)."""
    from aqualora_amd import metrics, synth
    dec = synthetic_decoder(48, device, seed).eval()
    x = (synth.normal("ba.img", (n_images, 3, 512, 512), 0.5, seed, device)).clamp(-1, 1)
    with torch.no_grad():
        logits = dec(x)
    bits = metrics.extract_bits(logits)
    obits, ologits = oracle_bits(dec, x)
    eq = (bits.cpu() == obits)
    rel = float((logits.float().cpu() - ologits).abs().max() / ologits.abs().max())
    margin = float((ologits[..., 1] - ologits[..., 0]).abs().min() / ologits.abs().max())
    acc = metrics.bit_accuracy(bits.cpu(), obits)     # utils_eval.py:199-203 with the oracle's decode as msg_gt
    rates = {}
    for nb in (1, 16):
        xb = x[:1].expand(nb, -1, -1, -1).contiguous()
        with torch.no_grad():
            ms, how = time_kernel(lambda: dec(xb), iters=5)
        rates[f"batch{nb}"] = {"images_per_sec": nb / ms * 1e3, "ms": ms}
    return {"bits_equal_to_oracle": float(eq.float().mean()), "n_bits": int(eq.numel()), "n_images": n_images,
            "bit_accuracy_vs_oracle_decode": float(acc.mean()), "logits_max_rel_err": rel, "smallest_logit_margin_rel": margin,
            "checker": "oracle/decoder_oracle.py (CPU restatement of utils/models.py:91-96 over torchvision's B1; unpinned)",
            "extract": rates, "timing": how, "weights": "synthetic (He-normal), eval mode, 48-bit head"}


def infer_record(device, B=1, runs=2, check_bits=True):
    """BASELINE config 4 (evaluation/run_eval_base.py:39-66): 50-step DDIM, CFG 7.5, 64x64x4 latents on the fused-LoRA U-Net (the
    LoRA is folded into W, utils_eval.py:81-82, so this is the plain SD-1.5 U-Net on batch 2 per image) -> frozen VAE decode to
    512x512 -> SecretDecoder bit extraction (utils_eval.py:131-140), all on the HIP kernels; the extracted bits are then checked
    against the CPU oracle's decode of the SAME decoded images."""
    from aqualora_amd import metrics, synth
    from aqualora_amd.inference import ddim_sample
    from aqualora_amd.unet import UNet2DConditionModel, init_synthetic
    from aqualora_amd.vae import SD15_VAE, AutoencoderKL, synthetic_state_dict
    unet = UNet2DConditionModel(device=device, dtype=torch.bfloat16)
    init_synthetic(unet, 2048)
    vae = AutoencoderKL(synthetic_state_dict(SD15_VAE, device=device), SD15_VAE, device)
    dec = synthetic_decoder(48, device).eval()
    ctx = synth.normal("inf.ctx", (B, 77, 768), 1.0, 1, device)
    lat = synth.normal("inf.lat", (B, 4, 64, 64), 1.0, 1, device)

    def pipeline():
        z = ddim_sample(unet, ctx, torch.zeros_like(ctx), lat, 50, 7.5)
        img = vae.decode(z.clamp(-4, 4) * 0.18215)      # synthetic weights: keep the latents in the VAE's range
        with torch.no_grad():
            bits = metrics.extract_bits(dec(img.clamp(-1, 1)))
        return z, img, bits

    pipeline()  # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(runs):
        z = ddim_sample(unet, ctx, torch.zeros_like(ctx), lat, 50, 7.5)
    torch.cuda.synchronize()
    dt_s = (time.perf_counter() - t0) / runs
    t0 = time.perf_counter()
    for _ in range(runs):
        z, img, bits = pipeline()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / runs
    tf = B * 50 * 2 * UNET_FWD_GFLOP / 1e3
    rec = {"workload": f"BASELINE config 4: batch {B}, 50 DDIM steps (ONE captured step graph kept across calls; timestep head and text k|v outside the step), CFG 7.5, "
                       "VAE decode to 512x512, 48-bit extraction; synthetic weights",
           "value": B / dt, "unit": "images/sec", "ms_per_image": 1e3 * dt / B, "runs": runs,
           "sampling_only": {"images_per_sec": B / dt_s, "ms_per_image": 1e3 * dt_s / B},
           "roofline": {"bound": "mfma", "achieved": tf / dt_s, "peak": MFMA_PEAK_TF, "unit": "TFLOP/s",
                        "frac": tf / dt_s / MFMA_PEAK_TF,
                        "launch": f"the 100 U-Net forwards of one image's sampling loop = {tf / B:.1f} TFLOP (SURVEY 8(d))"},
           "finite": bool(torch.isfinite(img).all())}
    if check_bits:
        obits, ologits = oracle_bits(dec, img.clamp(-1, 1))
        eq = bits.cpu() == obits
        rec["bits_equal_to_oracle"] = float(eq.float().mean())
        rec["n_bits"] = int(eq.numel())
        rec["smallest_logit_margin_rel"] = float((ologits[..., 1] - ologits[..., 0]).abs().min() / ologits.abs().max())
    del unet, vae, dec
    torch.cuda.empty_cache()
    return rec


def infer_bench(args, device):
    rec = infer_record(device, args.infer_batch, max(1, args.steps // 5))
    line = {"metric": "50-step DDIM txt2img + VAE decode + SecretDecoder extract, images/sec at 512x512", "n_gpus": 1,
            "steps": rec["runs"], "dtype": "bf16", "higher_is_better": True, "data": "synthetic",
            "config": {"workload": rec.pop("workload")}}
    line.update(rec)
    print(json.dumps(line), flush=True)


def robft_bench(args, device):
    """BASELINE config 5, the part that trains: rob_enhance_finetune.py:1018-1036 -- generated images (synthetic here)
    -> distortion -> SecretDecoder in train() mode (EfficientNet-B1, BatchNorm batch statistics) forward + backward ->
    BCE -> AdamW, batch 16 at 512x512.  The 20-step sampling pipeline that produces the images runs under no_grad and is
    `--mode infer`'s kernel path."""
    from aqualora_amd import noise as NZ, stage1 as S1, synth
    B = 16
    dec = synthetic_decoder(48, device).train()
    opt = torch.optim.AdamW(dec.parameters(), lr=1e-4)
    imgs = (synth.normal("rob.img", (B, 3, 512, 512), 0.25, 1, device) + 0.5).clamp(0, 1)
    bits = synth.bits("rob.bits", (B, 48), 1).to(device)
    distort = NZ.RobNoiser([0.6, 0.1, 0.15, 0.05, 0.1])
    gen = None
    drawn = []
    if args.robft_sample:
        # the whole iteration of rob_enhance_finetune.py:997-1036: a fresh random message per image -> S = mapper(m) * 1.03,
        # concatenated for the two CFG halves (:999-1002) -> 20-step DPM-Solver++ sampling (CFG 7.5) through the UN-fused
        # rank-r watermark LoRA at a random resolution from {512..768}^2 (:1004-1005) -> frozen VAE decode -> 8-bit
        # quantised [0,1] images (:1015-1021) -> distortion -> decoder training step on THOSE messages
        import random
        from aqualora_amd.inference import dpm_solver_sample
        from aqualora_amd.lora import inject_lora, patch_lora_forwards
        from aqualora_amd.unet import UNet2DConditionModel, init_synthetic, lora_keys
        from aqualora_amd.vae import SD15_VAE, AutoencoderKL, synthetic_state_dict
        from aqualora_amd.watermark import MapperNet
        r = args.rank
        unet = UNet2DConditionModel(device=device, dtype=torch.bfloat16)
        init_synthetic(unet, 2048)
        keys = lora_keys(unet)
        inject_lora(unet, r, keys)
        with torch.no_grad():
            for k in keys:
                lay = unet.get_submodule(k).lora_layer
                lay.down.weight.copy_(synth.normal(k + ".lora.down", lay.down.weight.shape, 1.0 / r, 2048, device))
                lay.up.weight.copy_(synth.normal(k + ".lora.up", lay.up.weight.shape, 0.02, 2048, device))
        patch_lora_forwards(unet)
        mapper = MapperNet(48, r).to(device)
        vae = AutoencoderKL(synthetic_state_dict(SD15_VAE, device=device), SD15_VAE, device)
        ctx = synth.normal("rob.ctx", (B, 77, 768), 1.0, 1, device)
        rng = random.Random(2048)
        sizes = [512, 576, 640, 704, 768] if args.robft_res == "random" else [int(args.robft_res)]
        it = [0]
        drawn = []

        def gen():
            it[0] += 1
            bits.copy_(synth.bits(f"rob.bits{it[0]}", (B, 48), 1).to(device))
            with torch.no_grad():
                S = mapper(bits.float()).to(torch.bfloat16) * 1.03
            h, w = rng.choice(sizes) // 8, rng.choice(sizes) // 8
            drawn.append((h * 8, w * 8))
            lat = synth.normal(f"rob.lat{it[0]}", (B, 4, h, w), 1.0, 1, device)
            z = dpm_solver_sample(unet, ctx, torch.zeros_like(ctx), lat, 20, 7.5, scale=S)
            img = (vae.decode(z.clamp(-4, 4) * 0.18215) / 2 + 0.5).clamp(0, 1)
            return torch.round(img.float() * 255.0) / 255.0

    def iteration():
        return S1.rob_finetune_step(dec, opt, gen() if gen is not None else imgs, bits, distort)

    for _ in range(max(1, args.warmup)):
        loss, acc = iteration()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, acc = iteration()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    gf = 3 * 6.4 * B  # fwd + bwd-data + bwd-weight of the 6.4 GFLOP/img network
    if getattr(args, "as_record", False):
        return {"workload": "BASELINE config 5, the part that trains (rob_enhance_finetune.py:1018-1036): distortion + SecretDecoder "
                            "(EfficientNet-B1, train mode) forward / backward / AdamW on given 512x512 images, batch 16, fp32",
                "value": B / dt, "unit": "images/sec", "ms_per_step": 1e3 * dt, "steps": args.steps, "dtype": "f32", "batch": B,
                "achieved_tflops_fp32": gf / dt / 1e3, "loss": float(loss), "bit_acc_of_step": float(acc),
                "finite": bool(torch.isfinite(loss)),
                **({} if gen is not None else {"roofline": (lambda nb, nbf, tr: {
                    "bound": "hbm", "achieved": nb / dt / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": nb / dt / 1e9 / 8000.0,
                    "traffic": None if tr is None else tr.get("traffic_bytes_per_step"),
                    "traffic_source": None if tr is None else "static, not measured in this run: " + tr.get("_how", "profiles/r06_pmc_robft.json"),
                    "bytes_per_step": nb,
                    "model": "algorithmic bytes of the decoder step counted op by op, fp32, no fusion between ops "
                             "(aqualora_amd.decoder.train_step_algorithmic_bytes: forward + backward-data + backward-weight of the "
                             "EfficientNet-B1 maps); the distortion layer and AdamW (26 MB of state) are not counted",
                    # the floor a FUSED step would have (BatchNorm-apply + SiLU read on the fly by the consumer, the squeeze-excite gate in
                    # the project conv's loader, the residual add in its epilogue): what `frac` would be against that model
                    "fused_model": {"bytes_per_step": nbf, "frac": nbf / dt / 1e9 / 8000.0,
                                    "model": "aqualora_amd.decoder.train_step_algorithmic_bytes(fused=True)"}})(
                                 __import__("aqualora_amd.decoder", fromlist=["x"]).train_step_algorithmic_bytes(B, 512),
                                 __import__("aqualora_amd.decoder", fromlist=["x"]).train_step_algorithmic_bytes(B, 512, fused=True),
                                 load_profile_json("r06_pmc_robft.json"))}),
                **({"resolutions_drawn": drawn} if gen is not None else {}),
                **({"generator": f"per-image messages, 20-step DPM-Solver++ sampling (CFG 7.5) through the un-fused rank-{args.rank} "
                                 f"watermark LoRA at {args.robft_res}x{args.robft_res} + VAE decode in front of the step "
                                 "(rob_enhance_finetune.py:997-1021): the WHOLE iteration"} if gen is not None else {})}
    print(json.dumps({"metric": "rob-finetune decoder step images/sec at 512x512 (EfficientNet-B1 train mode, fp32)",
                      "value": B / dt, "unit": "images/sec", "n_gpus": 1, "steps": args.steps, "ms_per_step": 1e3 * dt,
                      "dtype": "f32", "higher_is_better": True, "data": "synthetic", "batch": B,
                      "config": {"workload": f"per-image messages, 20-step DPM-Solver++ sampling through the un-fused rank-{args.rank} LoRA "
                                             f"(res {args.robft_res}) + VAE decode + distortion + decoder fwd/bwd/AdamW"
                                 if gen is not None else "distortion + decoder fwd/bwd/AdamW on given images"},
                      "achieved_tflops_fp32": gf / dt / 1e3, "loss": float(loss), "bit_acc": float(acc)}), flush=True)


def wrap_pixel_text(runner, args, device, rank_id, pixel, text):
    """The reference's own step starts from pixels and token ids (ppft_train.py:993, 1014-1019): optionally put the frozen
    VAE encode and / or the frozen CLIP text encoder (both on the HIP kernels) in front of the captured latent-in step."""
    if pixel:
        from aqualora_amd import synth
        from aqualora_amd.vae import SD15_VAE, AutoencoderKL, synthetic_state_dict
        vae = AutoencoderKL(synthetic_state_dict(SD15_VAE, device=device), SD15_VAE, device)
        pixels = synth.normal("bench.px", (args.batch, 3, 512, 512), 0.5, 2048 + 977 * rank_id, device).clamp(-1, 1)
        vnoise = synth.normal("bench.vn", (args.batch, 4, 64, 64), 1.0, 2048 + 977 * rank_id, device)
        latent_runner = runner

        def runner(**b):
            b["z"] = vae.encode(pixels, vnoise)
            return latent_runner(**b)

        runner.is_graph = getattr(latent_runner, "is_graph", False)
    if text:
        from aqualora_amd import synth
        from aqualora_amd.clip import SD15_CLIP, CLIPTextModel, clip_keys
        csd = {}
        for k, shp in clip_keys().items():
            if "layer_norm" in k:
                csd[k] = torch.ones(shp, device=device) if k.endswith("weight") else torch.zeros(shp, device=device)
            else:
                csd[k] = synth.normal("clip." + k, shp, 0.02 if k.endswith("bias") else (0.5 if "embedding" in k else shp[-1] ** -0.5),
                                      2048, device)
        clip = CLIPTextModel(csd, SD15_CLIP, device)
        ids = synth.randint("bench.ids", (args.batch, 77), 49408, 2048 + 977 * rank_id, device)
        prev_runner = runner

        def runner(**b):
            b["ctx"] = clip(ids)
            return prev_runner(**b)

        runner.is_graph = getattr(prev_runner, "is_graph", False)
    return runner


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=2, choices=[2, 3],
                    help="BASELINE.json config: 2 = rank 32, batch 4/GPU (the headline, 1 GPU); 3 = rank 320, batch 8/GPU "
                         "(the 8-GPU DDP recipe, train/README.md:34-48); --rank / --batch override")
    ap.add_argument("--rank", type=int, default=None)
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--no-extras", action="store_true", help="skip the pixel+text-in timing (VAE encode + CLIP inside the step)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--robft-sample", action="store_true",
                    help="robft mode: generate the batch inside the timed iteration (20-step DPM-Solver++, VAE decode)")
    ap.add_argument("--robft-res", default="random", help="robft --robft-sample: 'random' ({512..768}^2 like the reference) or a size")
    ap.add_argument("--infer-batch", type=int, default=1, help="infer mode: images sampled together")
    ap.add_argument("--text-in", action="store_true",
                    help="train mode: run the frozen CLIP text encoder (ids [B,77] -> [B,77,768]) inside every timed step")
    ap.add_argument("--pixel-in", action="store_true",
                    help="train mode: run the frozen VAE encode (3x512x512 -> 4x64x64) inside every timed step")
    ap.add_argument("--mode", choices=["train", "infer", "robft", "vae"], default="train",
                    help="train: the PPFT step (BASELINE metric); infer: 50-step DDIM + CFG latent sampling (config 4)")
    ap.add_argument("--micro", type=int, default=1, help="concurrent micro-batch slices per step")
    args = ap.parse_args()
    preset = {2: (32, 4), 3: (320, 8)}[args.config]
    args.rank = preset[0] if args.rank is None else args.rank
    args.batch = preset[1] if args.batch is None else args.batch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank_id = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    launched = "RANK" in os.environ  # under torch.distributed.run (also for N=1, so the RCCL path is the one measured)
    rccl_ranks_seen = 1
    if launched:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)
        one = torch.ones(1, device=device)
        dist.all_reduce(one)           # a real RCCL collective: every rank that is really there adds 1
        rccl_ranks_seen = int(one.item())

    if args.mode == "infer":
        return infer_bench(args, device)
    if args.mode == "vae":
        return vae_bench(args, device)
    if args.mode == "robft":
        return robft_bench(args, device)
    tr = build(device, args.rank, micro=args.micro)
    batch = synthetic_batch(args.batch, device, rank_id)
    runner = tr.step
    if not args.no_graph and hasattr(tr, "capture"):
        runner = tr.capture(batch)

    def barrier():
        if launched:
            dist.barrier()
        torch.cuda.synchronize()

    runner = wrap_pixel_text(runner, args, device, rank_id, args.pixel_in, args.text_in)

    # every step gets its own batch (fresh z / msg / eps / t / ctx), all generated up front so that the timed region
    # only contains device-to-device copies into the captured step's static buffers
    nset = min(args.steps, 16)
    batches = [batch] + [synthetic_batch(args.batch, device, rank_id, step=i) for i in range(1, max(nset, 1))]
    for i in range(args.warmup):
        loss = runner(**batches[i % len(batches)])
    barrier()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        evs[i][0].record()            # torch's current stream == the stream every C-ABI launch and graph replay uses
        loss = runner(**batches[i % len(batches)])
        evs[i][1].record()
    barrier()
    dt = time.perf_counter() - t0
    per_step = sorted(a.elapsed_time(b) for a, b in evs)
    if launched:
        tt = torch.tensor([dt], device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = tt.item()
    images = args.batch * world * args.steps
    value = images / dt
    loss_v = float(loss)
    exch = exchange_record(tr, device, world) if launched else None   # collective: every rank calls it

    if rank_id == 0:
        tf_img = step_tflop_per_image(args.rank)
        achieved = tf_img * args.batch / (dt / args.steps)  # TFLOP/s per GPU
        line = {
            "metric": "PPFT train-step images/sec at 512x512", "value": value, "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "ms_per_step_hip_event_median": per_step[len(per_step) // 2], "ms_per_step_hip_event_min": per_step[0],
            "rccl_ranks_seen": rccl_ranks_seen, "exchange_alone": exch,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"SD1.5 PPFT LoRA rank={args.rank}, 48-bit msg, 512x512 (64x64x4 latents in), "
                                   f"batch={args.batch}/GPU, " + ("pixel-in (frozen VAE encode inside the step" + (", CLIP text encoder inside" if args.text_in else ", CLIP outside") + ")"
                                                                  if args.pixel_in else "latent-in (VAE/CLIP outside the path)"),
                       "global_batch": args.batch * world, "parallelism": f"dp{world}", "baseline_config": args.config, "inputs": "fresh random batch every step", "hip_graph": bool(getattr(runner, "is_graph", False)),
                       "graphs_per_step": getattr(runner, "n_graphs", None),
                       "exchange": ("aql_comm_* (RCCL) on a forked stream, captured in the step graph; early buckets overlap backward"
                                    if tr.overlap else ("torch.distributed all-reduce between graphs (" + tr.comm_note + ")"
                                                        if world > 1 or os.environ.get("AQL_FORCE_ALLREDUCE") else "none (single GPU)")),
                       "loss": loss_v},
        }
        with torch.no_grad():
            ks = dominant_kernels(args.batch, device)
        dom = ks[0]
        # `roofline` = the heaviest launch of the TIME-dominant kernel family (the one-launch LoRA linears, `families` below),
        # timed live with HIP events on the launch stream.  `traffic` = HBM bytes per launch of the same kernel and shape from
        # the rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in separate passes, gfx950 2x FETCH correction, tools/pmc_traffic.sh),
        # read from the committed profiles/r06_pmc_traffic.json -- NOT measured in this run; null without a committed pass.
        pmc = load_profile_json("r06_pmc_traffic.json") or {}
        ent = pmc.get(dom.get("pmc_key", ""))
        traffic = None if ent is None else ent["traffic_bytes"]
        line["roofline"] = {"bound": "mfma", "kernel": dom["kernel"], "achieved": dom["achieved_tflops"],
                            "peak": MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": dom["frac_of_mfma_peak"],
                            "traffic": traffic, "traffic_unit": f"bytes/launch (algorithmic: {dom['algorithmic_bytes']:.4g})",
                            "traffic_source": None if traffic is None else
                            "static, not measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this kernel + shape, "
                            "profiles/r06_pmc_traffic.json",
                            "flops_per_launch": dom["flops"], "ms_per_launch": dom["ms"], "timing": dom["timing"],
                            "selected_by": "largest ms/step family of the committed kernel trace (families), heaviest launch of it",
                            "hbm_view": {"achieved_GBps": dom["algorithmic_bytes"] / dom["ms"] / 1e6, "peak_GBps": 8000.0,
                                         "frac": dom["algorithmic_bytes"] / dom["ms"] / 1e6 / 8000.0}}
        for k in ks:
            e = pmc.get(k.get("pmc_key", ""))
            if e is not None:
                k["traffic_bytes_static"] = e["traffic_bytes"]
        fam = load_profile_json(f"r06_families_config{args.config}.json")
        if fam is not None:
            line["families"] = {"source": f"static, not measured in this run: profiles/r06_families_config{args.config}.json "
                                          "(tools/prof_families.py over a rocprofv3 --kernel-trace of this command)",
                                "ms_per_step_profiled": fam.get("ms_per_step"), "launches_per_step": fam.get("launches_per_step"),
                                "rows": fam.get("families")}
        line["step_roofline"] = {"bound": "mfma", "achieved": achieved, "peak": MFMA_PEAK_TF, "unit": "TFLOP/s",
                                 "frac": achieved / MFMA_PEAK_TF,
                                 "launch": f"one PPFT step = {tf_img:.3f} TFLOP/image x {args.batch} images"}
        line["kernels"] = ks
        line["static_profile_sections"] = dict(profile_staleness(), sections=["roofline.traffic", "families", "mfma_util_pmc",
                                                                             "kernels[].traffic_bytes_static"])
        mu = load_profile_json("r06_pmc_mfma_util.json")
        if mu is not None:   # rocprofv3 MfmaUtil (matrix-pipe busy fraction) of the attention / conv / LoRA kernels: static evidence
            line["mfma_util_pmc"] = {"source": "static, not measured in this run: profiles/r06_pmc_mfma_util.json (rocprofv3 --pmc pass)",
                                     "kernels": {k: v["mfma_util"] for k, v in mu["kernels"].items()},
                                     "attention_64x64_time_weighted": mu.get("attention_64x64_time_weighted"),
                                     # SURVEY 8(d)'s subset "attention linears + SDPA" at the 64 x 64 level (formula in the file)
                                     "attention_linears_plus_sdpa_time_weighted": mu.get("attention_linears_plus_sdpa_time_weighted")}
        if world == 1 and not args.no_extras and not (args.pixel_in or args.text_in):
            # the reference's real step boundary: pixels and token ids in (frozen VAE encode + CLIP text encoder inside)
            full = wrap_pixel_text(runner, args, device, rank_id, True, True)
            n_x = max(3, min(args.steps, 10))
            for i in range(2):
                full(**batches[i % len(batches)])
            torch.cuda.synchronize()
            tx = time.perf_counter()
            for i in range(n_x):
                full(**batches[i % len(batches)])
            torch.cuda.synchronize()
            dtx = (time.perf_counter() - tx) / n_x
            line["pixel_text_in"] = {"value": args.batch / dtx, "unit": "images/sec", "ms_per_step": 1e3 * dtx, "steps": n_x,
                                     "workload": "same step with the frozen VAE encode (3x512x512 -> 4x64x64) and CLIP text "
                                                 "encoder (ids [B,77] -> [B,77,768]) inside, ppft_train.py:993,1014-1019"}
        if world == 1 and not args.no_extras and args.config == 2 and args.rank == 32 and not (args.pixel_in or args.text_in):
            line["config3"] = config3_record(device, rank_id)
        if world == 1 and not args.no_extras and args.config == 2 and args.rank == 32 and not (args.pixel_in or args.text_in):
            # the other half of BASELINE.json's metric and its configs 4 / 5, as sub-records of the driver's line
            import copy
            line["bit_accuracy"] = bit_accuracy_record(device)
            line["config4"] = infer_record(device, 1, runs=2)
            ra = copy.copy(args)
            ra.as_record, ra.steps, ra.warmup, ra.robft_sample = True, 10, 3, False
            line["config5"] = robft_bench(ra, device)
            # ... and the whole iteration of the reference's loop at BASELINE's rank 320: the 20-step sampling of the batch's 16 images
            # through the un-fused LoRA + VAE decode in front of the decoder step (the sampling is ~97 % of it)
            rb = copy.copy(args)
            # ... at the reference's resolutions: height and width drawn per step from {512, 576, 640, 704, 768} (:1004-1005), so the
            # timed steps include the capture of the sampling loop for every (height, width) pair seen for the first time
            rb.as_record, rb.steps, rb.warmup, rb.robft_sample, rb.robft_res, rb.rank = True, 2, 1, True, "random", 320
            whole = robft_bench(rb, device)
            line["config5"]["whole_iteration"] = {k: whole[k] for k in ("value", "unit", "ms_per_step", "steps", "generator", "finite",
                                                                         "resolutions_drawn")}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(tr, args.rank)
        print(json.dumps(line), flush=True)
    if launched:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
