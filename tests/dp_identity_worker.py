"""GPU worker (``python -m tests.dp_identity_worker RANK BATCH``, run by test_gpu_parity): the DEFAULT data-parallel form of the step
at FULL size -- the captured, hook-driven aql_comm_* exchange (three backward legs forked onto the side stream, ONE step graph;
dp.make_comm, the default since round 6) on a single-rank RCCL communicator (AQL_FORCE_ALLREDUCE=1) -- against the un-exchanged
single-GPU step from the same parameters and inputs.  The mean over ONE rank is the identity, so the two forms differ only in how
the weight-gradient launches are grouped (fp32 atomics: order-dependent in the last bits): the first loss must be BIT-equal (the
forward pass is deterministic), the exchanged gradient buffer and the parameters after two optimizer steps equal up to that order.
Reference: DDP's reducer fired from backward, train/ppft_train.py:905-912,1054-1058.  Prints one JSON line."""
import json
import os
import sys

import torch
import torch.distributed as dist


def main(rank, B):
    from aqualora_amd import synth
    from aqualora_amd.lora import inject_lora
    from aqualora_amd.ppft import PPFTTrainer
    from aqualora_amd.unet import UNet2DConditionModel, init_synthetic, lora_keys
    from aqualora_amd.watermark import MapperNet, SecretEncoder
    dev = "cuda"
    seed = 6006
    z = synth.normal("id.z", (B, 4, 64, 64), 1.0, seed).to(dev)
    wm = synth.normal("id.wm", (B, 4, 64, 64), 0.05, seed).to(dev)
    eps = synth.normal("id.eps", (B, 4, 64, 64), 1.0, seed).to(dev)
    msg = synth.bits("id.msg", (B, 48), seed).to(dev)
    ctx = synth.normal("id.ctx", (B, 77, 768), 1.0, seed).to(dev).to(torch.bfloat16)
    t = torch.tensor([500, 20, 981, 333, 7, 760, 129, 611][:B], device=dev)
    batch = dict(z=z, msg=msg, eps=eps, t=t, ctx=ctx)
    unet = UNet2DConditionModel(device=dev, dtype=torch.bfloat16)
    init_synthetic(unet, seed)
    keys = lora_keys(unet)
    inject_lora(unet, rank, keys)
    with torch.no_grad():
        for k in keys:
            lay = unet.get_submodule(k).lora_layer
            lay.down.weight.copy_(synth.normal(k + ".lora.down", lay.down.weight.shape, 1.0 / rank, seed, dev))
            lay.up.weight.copy_(synth.normal(k + ".lora.up", lay.up.weight.shape, 0.02, seed, dev))
    out, res, flat0 = {"rank": rank, "batch": B}, {}, None
    for mode in ("plain", "exchange"):
        os.environ.pop("AQL_FORCE_ALLREDUCE", None)
        os.environ.pop("AQL_COMM", None)          # unset: the default must BE the captured exchange
        if mode == "exchange":
            os.environ["AQL_FORCE_ALLREDUCE"] = "1"
        mapper = MapperNet(48, rank)
        with torch.no_grad():
            mapper.bit_embeddings.weight.copy_(synth.normal("id.E", (48, rank), 1.0, seed))
        tr = PPFTTrainer(unet, mapper, SecretEncoder(48), rank)      # (the U-Net keeps its LoRA layers: same parameters both times)
        tr.sec_encoder.encode = lambda m, out_scale=1.0: wm
        if flat0 is None:
            flat0 = tr.bank.flat.clone()
        else:
            tr.bank.flat.copy_(flat0)
            tr.bank.refresh()
        run = tr.capture(batch, warmup=0)
        losses = [float(run(**batch))]
        torch.cuda.synchronize()
        # AdamW's first moment after the FIRST step = (1 - beta1) x the (exchanged, clipped) gradients of that step, taken from the same
        # parameters in both forms (the step zeroes the gradient buffer itself)
        g1 = tr.bank.exp_avg[:tr.bank.numel].clone()
        losses.append(float(run(**batch)))
        torch.cuda.synchronize()
        res[mode] = (losses, tr.bank.flat.clone(), g1)
        out[f"{mode}_overlap"], out[f"{mode}_n_graphs"] = bool(tr.overlap), int(getattr(run, "n_graphs", 0))
        out[f"{mode}_comm_note"] = tr.comm_note
        if mode == "exchange":
            out["exchange_ranges"] = [list(map(int, r)) for r in list(tr.early_ranges) + list(tr.late_ranges)]
            out["numel"] = int(tr.bank.numel)
        del tr, run
    (lp, pp, gp), (le, pe, ge) = res["plain"], res["exchange"]
    out.update(plain_losses=lp, exchange_losses=le, first_loss_bit_equal=bool(lp[0] == le[0]),
               grad_relerr=float((ge - gp).norm() / gp.norm()), param_relerr=float((pe - pp).abs().max() / pp.abs().max()),
               finite=bool(torch.isfinite(pe).all() and torch.isfinite(ge).all()))
    print(json.dumps(out))


if __name__ == "__main__":
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29551")
    dist.init_process_group("nccl", rank=0, world_size=1)
    torch.cuda.set_device(0)
    try:
        main(int(sys.argv[1]), int(sys.argv[2]))
    finally:
        dist.destroy_process_group()
    sys.exit(0)
