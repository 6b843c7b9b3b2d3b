"""GPU worker (``python -m tests.dp_config3_worker``, run by test_gpu_parity): BASELINE config 3 per GPU -- rank 320, batch 8 --
end to end at FULL size through the data-parallel form of the step.  A single-rank RCCL group with AQL_FORCE_ALLREDUCE=1 puts
the trainer into the form every rank of the 8-GPU recipe runs (train/README.md:34-48, ppft_train.py:1058): the wide-rank
weight-gradient GEMMs are held back and cut into exchange buckets of the 543 MB gradient buffer -- by default the overlapped
exchange through aql_comm_* (up-path buckets forked from the backward hook, the rest behind the last weight-gradient launch,
captured into ONE step graph); with AQL_COMM=0 the torch.distributed form (8 bucket graphs, eager collectives).

Samples are independent, so ONE batch-8 twin step must equal the mean of eight batch-1 steps (the batch-1 step at rank 320 is
pinned to the CPU oracle by test_full_size_ppft_gradients_vs_oracle[320]).  Shapes that only exist here: conv_row_kernel<32,8>,
two-round 256x160 grids, aql_gemm_bf16_geglu_bwd at M = 32768, 65536-row twin GEMMs, the bucketed wide-rank run_bucket path.
Prints one JSON line."""
import json
import os
import sys

import torch
import torch.distributed as dist


def l2rel(a, b):
    a, b = a.float(), b.float()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def main():
    from aqualora_amd import synth
    from aqualora_amd.lora import inject_lora
    from aqualora_amd.ppft import PPFTTrainer
    from aqualora_amd.unet import UNet2DConditionModel, init_synthetic, lora_keys
    from aqualora_amd.watermark import MapperNet, SecretEncoder
    dev = "cuda"
    seed, rank, B = 4096, 320, 8
    unet = UNet2DConditionModel(device=dev, dtype=torch.bfloat16)
    init_synthetic(unet, seed)
    keys = lora_keys(unet)
    inject_lora(unet, rank, keys)
    with torch.no_grad():
        for k in keys:
            lay = unet.get_submodule(k).lora_layer
            lay.down.weight.copy_(synth.normal(k + ".lora.down", lay.down.weight.shape, 1.0 / rank, seed, dev))
            lay.up.weight.copy_(synth.normal(k + ".lora.up", lay.up.weight.shape, 0.02, seed, dev))
    mapper = MapperNet(48, rank)
    with torch.no_grad():
        mapper.bit_embeddings.weight.copy_(synth.normal("c3.E", (48, rank), 1.0, seed))
    tr = PPFTTrainer(unet, mapper, SecretEncoder(48), rank)
    out = {"bucketed": bool(tr.bucketed), "overlap": bool(tr.overlap), "comm": tr.comm_note, "twin": bool(tr.twin),
           "n_lora": int(tr.bank.n_lora), "n_early": int(tr.bank.n_early), "numel": int(tr.bank.numel),
           "cuts": [int(c) for c in tr.bank.cuts], "n_legs": int(getattr(tr, "n_legs", 0))}
    z = synth.normal("c3.z", (B, 4, 64, 64), 1.0, seed).to(dev)
    wm = synth.normal("c3.wm", (B, 4, 64, 64), 0.05, seed).to(dev)
    eps = synth.normal("c3.eps", (B, 4, 64, 64), 1.0, seed).to(dev)
    msg = synth.bits("c3.msg", (B, 48), seed).to(dev)
    ctx = synth.normal("c3.ctx", (B, 77, 768), 1.0, seed).to(dev).to(torch.bfloat16)
    t = torch.tensor([500, 20, 981, 333, 7, 760, 129, 611], device=dev)
    cur = {"sl": slice(0, B)}
    tr.sec_encoder.encode = lambda m, out_scale=1.0: wm[cur["sl"]]
    n = tr.bank.numel

    def one(sl):
        """forward + backward + the gradient exchange in the trainer's data-parallel form: the overlapped exchange through
        aql_comm_* (early buckets from the backward hook, late buckets at the end), or -- AQL_COMM=0 -- weight gradients held
        back, then launch bucket k / all-reduce it through torch.distributed"""
        cur["sl"] = sl
        tr.bank.zero_grad()
        if tr.overlap:
            kinds = set()
            tables = list(tr.deferred_legs) + [tr.deferred]
            real = [d.plan for d in tables]
            for d, f in zip(tables, real):
                d.plan = (lambda *a, _d=d, _f=f: (kinds.update(it[1] for it in _d.items), _f(*a))[1])
            loss, pred, clean = tr.forward_backward(z[sl], msg[sl], eps[sl], t[sl], ctx[sl])
            for d, f in zip(tables, real):
                d.plan = f
            return loss, pred, clean, list(tr.early_ranges) + list(tr.late_ranges), sorted(kinds)
        loss, pred, clean = tr.forward_backward(z[sl], msg[sl], eps[sl], t[sl], ctx[sl], flush_dw=False)
        kinds = sorted({it[1] for it in tr.deferred.items})
        ranges = tr.plan_exchange()
        tr.exchange_bucketed(ranges, tr.deferred.run_bucket)
        tr.deferred.reset()
        return loss, pred, clean, ranges, kinds

    loss8, pred8, clean8, ranges, kinds = one(slice(0, B))
    g8 = tr.bank.grad[:n].clone()
    out["ranges"] = [list(map(int, r)) for r in ranges]
    out["problem_kinds"] = kinds            # "w" = wide-rank transpose-read problems of the grouped launch
    acc = torch.zeros_like(g8)
    losses, ep, ec = [], [], []
    for i in range(B):
        li, pi, ci, _, _ = one(slice(i, i + 1))
        acc += tr.bank.grad[:n] / B
        losses.append(float(li))
        ep.append(l2rel(pred8[i:i + 1], pi))
        ec.append(l2rel(clean8[i:i + 1], ci))
    torch.cuda.synchronize()
    out.update(loss8=float(loss8), loss_mean_b1=sum(losses) / B, pred_l2rel=ep, clean_l2rel=ec,
               grad_finite=bool(torch.isfinite(g8).all()), grad_l2rel=l2rel(g8, acc),
               grad_lora_l2rel=l2rel(g8[:tr.bank.n_lora], acc[:tr.bank.n_lora]),
               grad_mapper_l2rel=l2rel(g8[tr.bank.n_lora:], acc[tr.bank.n_lora:]), grad_norm=float(g8.norm()))
    # the captured form of the same step (what bench.py --config 3 under torchrun replays) against the eager form, from the
    # same parameters: two optimizer steps each
    batch = dict(z=z, msg=msg, eps=eps, t=t, ctx=ctx)
    cur["sl"] = slice(0, B)
    flat0 = tr.bank.flat.clone()

    def rewind():
        tr.bank.flat.copy_(flat0)
        tr.bank.exp_avg.zero_()
        tr.bank.exp_avg_sq.zero_()
        tr.step_t.zero_()
        tr.bank.zero_grad()
        tr.bank.refresh()

    rewind()
    le = [float(tr.step(**batch)) for _ in range(2)]
    p_eager = tr.bank.flat.clone()
    rewind()
    run = tr.capture(batch, warmup=0)
    lg = [float(run(**batch)) for _ in range(2)]
    torch.cuda.synchronize()
    out.update(eager_losses=le, graph_losses=lg,
               graph_vs_eager_param_relerr=float((tr.bank.flat - p_eager).abs().max() / p_eager.abs().max()),
               graph_buckets=len(tr.early_ranges) + len(tr.late_ranges) if tr.overlap else len(tr.exchange_ranges),
               n_graphs=int(getattr(run, "n_graphs", 0)), params_finite=bool(torch.isfinite(tr.bank.flat).all()))
    print(json.dumps(out))


if __name__ == "__main__":
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29549")
    os.environ["AQL_FORCE_ALLREDUCE"] = "1"
    os.environ.setdefault("AQL_COMM", "1")     # the overlapped aql_comm_* exchange (opt-in); AQL_COMM=0 tests the fallback
    dist.init_process_group("nccl", rank=0, world_size=1)
    torch.cuda.set_device(0)
    try:
        main()
    finally:
        dist.destroy_process_group()
    sys.exit(0)
