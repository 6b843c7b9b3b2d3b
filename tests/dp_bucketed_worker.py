"""GPU worker (run as ``python -m tests.dp_bucketed_worker`` by test_gpu_parity): a single-rank RCCL process group with
AQL_FORCE_ALLREDUCE=1 switches PPFTTrainer to its data-parallel form -- weight-gradient GEMMs in buckets, one
asynchronous all-reduce per bucket, bucket graphs between the forward/backward graph and the optimizer graph -- and the
result must equal the single-GPU form (one grouped launch, no collective) on the same inputs.  Prints one JSON line."""
import json
import os
import sys

import torch
import torch.distributed as dist


def build(rank, inp, lr=1e-3):
    from aqualora_amd.ppft import PPFTTrainer
    from aqualora_amd.watermark import MapperNet, SecretEncoder
    from tests.common import T
    from tests.test_gpu_parity import _gpu_tiny
    unet, keys, lw = _gpu_tiny(rank=rank)
    mapper = MapperNet(48, rank)
    with torch.no_grad():
        mapper.bit_embeddings.weight.copy_(inp["E"])
    enc = SecretEncoder(48, base_res=8, resolution=16)
    with torch.no_grad():
        enc.secret_scaler[0].weight.copy_(T("cap.lin.w", (64, 48), 48 ** -0.5))
        enc.secret_scaler[0].bias.copy_(T("cap.lin.b", (64,), 0.1))
        enc.secret_scaler[5].weight.copy_(T("enc.conv.w", (4, 4, 3, 3), 0.05))
    return PPFTTrainer(unet, mapper, enc, rank, learning_rate=lr)


def main():
    from tests.common import ppft_inputs
    dev = "cuda"
    out = {}
    for rank in (8, 40):   # 40 > 32: the weight gradients are "wide" problems, held back as direct launches
        inp = ppft_inputs(device=dev, rank=rank)
        batch = dict(z=inp["z"], msg=inp["msg"], eps=inp["eps"], t=inp["t"], ctx=inp["ctx"].to(torch.bfloat16))
        res = {}
        # plain: single-GPU form.  overlap_*: our RCCL communicator (aql_comm_*), collectives forked onto a side stream from the
        # backward hook and captured into the ONE step graph.  bucketed_*: the torch.distributed form (the default without AQL_COMM=1), bucket
        # graphs with eager collectives between them -- also the fallback when the communicator's self-test fails.
        for mode in ("plain", "overlap_eager", "overlap_graph", "bucketed_eager", "bucketed_graph"):
            os.environ.pop("AQL_FORCE_ALLREDUCE", None)
            os.environ.pop("AQL_COMM", None)
            os.environ.pop("AQL_BUCKETS", None)
            if mode != "plain":
                os.environ["AQL_FORCE_ALLREDUCE"] = "1"
                os.environ["AQL_BUCKETS"] = "3"     # the tiny bank is far below the size where bucketing switches on
            if mode != "plain":                     # the captured aql_comm_* exchange (the default) or, AQL_COMM=0, torch.distributed
                os.environ["AQL_COMM"] = "1" if mode.startswith("overlap") else "0"
            tr = build(rank, inp)
            assert tr.bucketed == mode.startswith("bucketed"), (mode, tr.bucketed, tr.comm_note)
            assert tr.overlap == mode.startswith("overlap"), (mode, tr.overlap, tr.comm_note)
            run = tr.capture(batch, warmup=0) if mode.endswith("_graph") else tr.step
            losses = [float(run(**batch)) for _ in range(3)]
            torch.cuda.synchronize()
            res[mode] = (losses, tr.bank.flat.clone())
            if mode == "bucketed_graph":
                out[f"r{rank}_ranges"] = [list(map(int, r)) for r in tr.exchange_ranges]
                out[f"r{rank}_n_lora"] = int(tr.bank.n_lora)
            if mode == "overlap_graph":
                out[f"r{rank}_overlap_ranges"] = [list(map(int, r)) for r in list(tr.early_ranges) + list(tr.late_ranges)]
                out[f"r{rank}_n_early"], out[f"r{rank}_numel"] = int(tr.bank.n_early), int(tr.bank.numel)
                out[f"r{rank}_cuts"] = [int(c) for c in tr.bank.cuts]
                out[f"r{rank}_hook_buckets"] = len(tr.early_ranges)
                out[f"r{rank}_overlap_graphs"] = int(getattr(run, "n_graphs", 0))
        lp, pp = res["plain"]
        for mode in ("overlap_eager", "overlap_graph", "bucketed_eager", "bucketed_graph"):
            l, p = res[mode]
            out[f"r{rank}_{mode}_param_relerr"] = float((p - pp).abs().max() / pp.abs().max())
            out[f"r{rank}_{mode}_losses"] = l
        out[f"r{rank}_plain_losses"] = lp
    out.update(robft_modes())
    print(json.dumps(out))


def robft_modes():
    """rob_enhance_finetune.py:1018-1036 on one GPU in its three exchange forms: no exchange; dp.ModuleGradExchange over
    torch.distributed (bucket all-reduces launched from gradient hooks during backward); the same through aql_comm_* on a forked
    side stream (AQL_COMM=1).  Two optimizer steps each from the same decoder, parameters compared with the plain form."""
    from aqualora_amd import stage1 as S1, synth
    from tests.test_gpu_parity import _synthetic_decoder
    B = 4
    img = (synth.normal("rob.dp.img", (B, 3, 128, 128), 0.25, 5, "cuda") + 0.5).clamp(0, 1)
    bits = synth.bits("rob.dp.bits", (B, 48), 5, "cuda")
    out, ref = {}, None
    for mode in ("plain", "plain2", "dist", "comm"):     # plain2: the run-to-run spread of the decoder step itself (fp32 atomics)
        os.environ.pop("AQL_FORCE_ALLREDUCE", None)
        os.environ.pop("AQL_COMM", None)
        if not mode.startswith("plain"):
            os.environ["AQL_FORCE_ALLREDUCE"] = "1"
        if not mode.startswith("plain"):
            os.environ["AQL_COMM"] = "1" if mode == "comm" else "0"
        torch.manual_seed(11)       # the train-mode forward draws stochastic depth / dropout from torch's generator
        dec = _synthetic_decoder(48).to("cuda").train()
        S1.prepare_rob_finetune(dec, None, n_buckets=4)
        ex = getattr(dec, "_aql_exchange", None)
        opt = torch.optim.AdamW(dec.parameters(), lr=1e-3)
        losses, orders = [], []
        for _ in range(2):
            if ex is not None:
                real = ex.finish
                ex.finish = lambda _r=real: (orders.append(_r()), orders[-1])[1]
            loss, acc = S1.rob_finetune_step(dec, opt, img, bits, None, None)
            losses.append(float(loss))
        torch.cuda.synchronize()
        flat = torch.cat([p.detach().reshape(-1) for p in dec.parameters()])
        if mode == "plain":
            ref = flat
            out["robft_plain_losses"] = losses
            assert ex is None
        elif mode == "plain2":
            out["robft_plain_rerun_param_relerr"] = float((flat - ref).abs().max() / ref.abs().max())
        else:
            assert ex is not None and (ex.comm is not None) == (mode == "comm"), (mode, ex.comm)
            out[f"robft_{mode}_param_relerr"] = float((flat - ref).abs().max() / ref.abs().max())
            out[f"robft_{mode}_losses"] = losses
            out[f"robft_{mode}_buckets"] = len(ex.ranges)
            out[f"robft_{mode}_hook_launch_order"] = orders[-1]
            out[f"robft_{mode}_bucket_mb"] = [round(4 * (hi - lo) / 2 ** 20, 2) for lo, hi in ex.ranges]
    os.environ.pop("AQL_FORCE_ALLREDUCE", None)
    os.environ.pop("AQL_COMM", None)
    return out


if __name__ == "__main__":
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group("nccl", rank=0, world_size=1)
    torch.cuda.set_device(0)
    try:
        main()
    finally:
        dist.destroy_process_group()
    sys.exit(0)
