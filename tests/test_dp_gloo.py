"""CPU, world_size 2 over gloo: the data-parallel gradient exchange (mean over ranks of the flat gradient buffer,
plain and bucketed) and rank-dependent synthetic batches."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from aqualora_amd import dp, synth
    n = 10007
    g = synth.normal("grad", (n,), 1.0, seed=100 + rank)
    want = sum(synth.normal("grad", (n,), 1.0, seed=100 + r) for r in range(world)) / world
    a = dp.allreduce_mean_(g.clone())
    b = dp.allreduce_mean_(g.clone(), bucket_elems=4096)
    z0 = synth.normal("bench.z", (2, 4, 8, 8), 1.0, 2048 + 977 * rank)
    gathered = [torch.zeros_like(z0) for _ in range(world)]
    dist.all_gather(gathered, z0)
    # the logged loss (ppft_train.py:1054): mean over ranks of the per-rank loss
    ok_log = abs(dp.logged_loss(torch.tensor(float(rank + 1))) - sum(range(1, world + 1)) / world) < 1e-6
    # bucketed asynchronous form used by the trainer: one collective per contiguous range, finished together
    c = g.clone()
    red = dp.BucketedAllreduce()
    for lo, hi in ((0, 3000), (3000, 3001), (3001, n)):
        red.launch(c[lo:hi])
    red.finish()
    ok_async = torch.allclose(c, want, atol=1e-6) and dp.exchange_active() and not red.pending and ok_log
    # DDP-style exchange for an ordinary module (the SecretDecoder of rob-finetune): bucketed flat gradient mean + buffers
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.BatchNorm1d(5), torch.nn.Linear(5, 3))
    x = synth.normal("dp.x", (6, 7), 1.0, seed=200 + rank)
    net(x).square().sum().backward()
    local = [p.grad.clone() for p in net.parameters()]
    gathered_g = [[torch.zeros_like(g) for _ in range(world)] for g in local]
    for g, out in zip(local, gathered_g):
        dist.all_gather(out, g)
    dp.allreduce_module_grads_(net.parameters(), bucket_bytes=64)      # tiny buckets: several collectives
    ok_mod = all(torch.allclose(p.grad, sum(o) / world, atol=1e-6) for p, o in zip(net.parameters(), gathered_g))
    net[1].running_mean.fill_(float(rank + 1))
    dp.broadcast_buffers_(net)
    ok_mod = ok_mod and float(net[1].running_mean[0]) == 1.0
    # construction-time parameter sync (DDP's broadcast from rank 0): ranks start from DIFFERENT draws (the reference
    # default is seed=None) and must hold rank 0's values afterwards -- the flat trainable buffer of the PPFT trainer ...
    flat = synth.normal("init.flat", (4099,), 1.0, seed=500 + rank)
    dp.broadcast_(flat)
    ok_mod = ok_mod and torch.equal(flat, synth.normal("init.flat", (4099,), 1.0, seed=500))
    # ... and an ordinary module (the rob-finetune decoder): parameters and buffers
    torch.manual_seed(1234 + rank)
    net2 = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.BatchNorm1d(5))
    net2[1].running_var.fill_(float(rank + 2))
    dp.broadcast_module_(net2)
    torch.manual_seed(1234)
    ref2 = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.BatchNorm1d(5))
    ok_mod = ok_mod and all(torch.equal(a_, b_) for a_, b_ in zip(net2.parameters(), ref2.parameters()))
    ok_mod = ok_mod and float(net2[1].running_var[0]) == 2.0
    # the overlapped exchange of PPFTTrainer (aql_comm_* on the GPU; the same protocol over gloo here): the buffer is
    # [early region | late region | mapper]; the early region is planned and reduced first (from the backward hook), the late
    # one at the end of backward with the mapper gradient riding on its last bucket -- together they must give the plain mean
    from aqualora_amd import ops
    sizes = [257, 1024, 33, 512, 4096, 7, 900, 2048, 129]                # nine weight-gradient outputs, back to back
    offs = [sum(sizes[:i]) for i in range(len(sizes))]
    n_lora, n_early = sum(sizes), sum(sizes[:4])
    d = synth.normal("grad", (n,), 1.0, seed=100 + rank)[:n_lora + 100].clone()   # + a 100-element "mapper" tail
    want_d = want[:n_lora + 100]
    # round 4: three backward legs (up path | mid + down_blocks.3/.2 | down_blocks.1) are planned and reduced from their own
    # hooks, in leg order; what is left (down_blocks.0, text projections) + the mapper tail at the end of backward
    legs = [(0, 4), (4, 6), (6, 7)]
    ranges = []
    for a_, b_ in legs:
        ranges += [(lo, hi) for lo, hi, _ in ops.plan_buckets(offs[a_:b_], sizes[a_:b_], 2)]
    ranges += [(lo, hi) for lo, hi, _ in ops.plan_buckets(offs[7:], sizes[7:], 3)]
    ranges[-1] = (ranges[-1][0], d.numel())
    cut_ends = [sum(sizes[:b_]) for _, b_ in legs]
    ok_tile = (ranges[0][0] == 0 and all(any(hi == c for _, hi in ranges) for c in cut_ends) and cut_ends[0] == n_early
               and all(x[1] == y[0] for x, y in zip(ranges, ranges[1:])))
    red2 = dp.BucketedAllreduce()
    for lo, hi in ranges:
        red2.launch(d[lo:hi])
    red2.finish()
    ok_async = ok_async and ok_tile and torch.allclose(d, want_d, atol=1e-6)
    # DDP's broadcast_buffers as ONE collective per dtype (fp32 statistics + int64 counters), and the hook-driven gradient
    # exchange of an ordinary module (dp.ModuleGradExchange: the rob-finetune decoder): mean of the ranks' gradients, every bucket
    # launched from a gradient hook during backward, the views survive zero_grad
    torch.manual_seed(77 + rank)
    net3 = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.BatchNorm1d(16), torch.nn.ReLU(), torch.nn.Linear(16, 8),
                               torch.nn.BatchNorm1d(8), torch.nn.Linear(8, 4))
    net3[1].running_mean.fill_(float(rank + 1))
    net3[4].num_batches_tracked.fill_(5 + rank)
    dp.broadcast_module_(net3)                       # parameters: rank 0's
    net3[1].running_mean.fill_(float(rank + 1))
    net3[4].num_batches_tracked.fill_(5 + rank)
    dp.broadcast_buffers_(net3)
    ok_buf = float(net3[1].running_mean[3]) == 1.0 and int(net3[4].num_batches_tracked) == 5
    ex = dp.ModuleGradExchange(net3, None, None, n_buckets=3)
    xin = synth.normal("mge.x", (5, 6), 1.0, seed=300 + rank)
    net3(xin).square().sum().backward()
    launched = ex.finish()
    mine = [p.grad.clone() for p in net3.parameters()]
    gath = [[torch.zeros_like(g_) for _ in range(world)] for g_ in mine]
    ok_mge = launched == list(range(len(ex.ranges))) and len(ex.ranges) >= 2
    # reference: plain local gradients (no exchange), averaged by hand
    ref3 = [p.detach().clone().requires_grad_(True) for p in net3.parameters()]
    import copy
    net4 = copy.deepcopy(net3)
    for p_ in net4.parameters():
        p_.grad = None
    for m_ in net4.modules():       # the first forward above already moved the running statistics: same batch statistics either way
        if isinstance(m_, torch.nn.BatchNorm1d):
            m_.momentum = 0.0
    net4(xin).square().sum().backward()
    for g_, p_ in zip(mine, net4.parameters()):
        loc = p_.grad.clone()
        dist.all_reduce(loc)
        ok_mge = ok_mge and torch.allclose(g_, loc / world, atol=1e-5, rtol=1e-5)
    ex.zero_grad()
    ok_mge = ok_mge and float(ex.flat.abs().sum()) == 0.0 and all(p_.grad is not None for p_ in net3.parameters())
    ok_mod = ok_mod and ok_buf and ok_mge
    # the trainer's choice of exchange (dp.make_comm, round 6: the captured aql_comm_* exchange is the DEFAULT under RCCL): under a
    # non-RCCL backend every rank gets (None, "backend gloo") -- the torch.distributed fallback -- and AQL_COMM=0 asks for it outright
    comm, note = dp.make_comm(None)
    os.environ["AQL_COMM"] = "0"
    comm0, note0 = dp.make_comm(None)
    del os.environ["AQL_COMM"]
    ok_mod = ok_mod and comm is None and note == "backend gloo" and comm0 is None and note0.startswith("AQL_COMM=0")
    q.put((rank, torch.allclose(a, want, atol=1e-6), torch.equal(a, b) and ok_async and ok_mod,
           not torch.equal(gathered[0], gathered[1])))
    dist.destroy_process_group()


def test_allreduce_mean_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(ok1 and ok2 and ok3 for _, ok1, ok2, ok3 in res), res


def test_plan_buckets_tiles_the_flat_buffer():
    """Host logic of the bucketed exchange: problems arrive in backward order (roughly, not exactly, the buffer order),
    buckets must be contiguous, disjoint, gap-free ranges that contain every problem's output exactly once."""
    import random
    from aqualora_amd.ops import plan_buckets
    from aqualora_amd import dp
    rng = random.Random(7)
    sizes = [rng.choice([320 * 8, 8 * 320, 2560 * 8, 8 * 1280, 640 * 8]) for _ in range(384)]
    offs, o = [], 0
    for s in sizes:
        offs.append(o)
        o += s
    perm = list(range(384))
    for i in range(0, 380, 4):           # local shuffles: q/k/v backward order is autograd's choice
        blk = perm[i:i + 4]
        rng.shuffle(blk)
        perm[i:i + 4] = blk
    p_offs, p_sizes = [offs[i] for i in perm], [sizes[i] for i in perm]
    for nb in (1, 2, 4, 8):
        plan = plan_buckets(p_offs, p_sizes, nb)
        assert 1 <= len(plan) <= nb
        assert plan[0][0] == 0 and plan[-1][1] == o
        seen = []
        for (lo, hi, items), nxt in zip(plan, plan[1:] + [None]):
            assert lo < hi and (nxt is None or nxt[0] == hi)
            for i in items:
                assert lo <= p_offs[i] and p_offs[i] + p_sizes[i] <= hi
            seen += items
        assert sorted(seen) == list(range(384))
        if nb > 1:
            tot = [sum(p_sizes[i] for i in items) for _, _, items in plan]
            assert max(tot) < 2.0 * o / len(plan)      # balanced
    assert plan_buckets([], [], 4) == []
    try:
        plan_buckets([0, 5], [10, 10], 2)
        raise AssertionError("overlap not detected")
    except ValueError:
        pass
    assert dp.bucket_count(54 << 20) == 1 and dp.bucket_count(543 << 20) == 8 and dp.bucket_count(200 << 20) == 3


def test_bank_order_puts_the_up_path_first_and_the_text_projections_last():
    """lora.bank_order (host logic of the flat gradient buffer's layout): reverse traversal, text-state k|v projections at the
    end, the up path a gap-free prefix -- what lets the overlapped exchange all-reduce the head of the buffer from the hook on
    the mid-block output while the mid / down backward is still running."""
    from aqualora_amd.lora import bank_order
    from aqualora_amd.unet import lora_keys
    from tests.common import tiny_unet
    keys = lora_keys(tiny_unet())
    order, n_lead = bank_order(keys)
    assert sorted(order) == list(range(len(keys)))
    names = [keys[i] for i in order]
    is_kv = lambda k: k.endswith(".attn2.to_k") or k.endswith(".attn2.to_v")   # noqa: E731
    n_kv = sum(is_kv(k) for k in keys)
    assert all(is_kv(k) for k in names[-n_kv:]) and not any(is_kv(k) for k in names[:-n_kv])
    assert n_lead == sum(k.startswith("up_blocks.") and not is_kv(k) for k in keys) > 0
    assert all(k.startswith("up_blocks.") for k in names[:n_lead]) and names[n_lead].startswith("mid_block.")
    body = names[:-n_kv]
    assert body == [k for k in reversed(keys) if not is_kv(k)]                 # gradient-ready order otherwise


def test_bank_stages_cut_the_buffer_at_the_backward_legs():
    """lora.bank_stages / LoraBank.cuts: the flat gradient buffer is [up path | mid + down_blocks.3/.2 | down_blocks.1 | rest];
    at rank 32 the grouped text-state projections sit at the end, at other ranks they stay inside their block.  Sizes at the
    SD-1.5 widths: what is left for the end of backward is 8.8 MB of 54 MB at rank 32 and 30 MB of 543 MB at rank 320."""
    from aqualora_amd.lora import backward_stage, bank_stages
    from aqualora_amd.unet import lora_keys
    from tests.common import tiny_unet
    keys = lora_keys(tiny_unet())
    is_kv = lambda k: k.endswith(".attn2.to_k") or k.endswith(".attn2.to_v")   # noqa: E731
    for kv_last in (True, False):
        order, ends = bank_stages(keys, kv_last)
        names = [keys[i] for i in order]
        assert sorted(order) == list(range(len(keys))) and len(ends) == 3 and 0 < ends[0] < ends[1] < ends[2] < len(keys)
        for q, e in enumerate(ends):
            assert all(backward_stage(k) <= q for k in names[:e])
            assert backward_stage(names[e]) > q or (kv_last and is_kv(names[e]))
        if kv_last:
            assert not any(is_kv(k) for k in names[:ends[2]])
        else:
            assert names == list(reversed(keys)) and any(is_kv(k) for k in names[:ends[0]])
    assert [backward_stage(k) for k in ("up_blocks.1.attentions.0.proj_in", "mid_block.attentions.0.proj_out",
                                        "down_blocks.2.attentions.1.proj_in", "down_blocks.1.attentions.0.proj_in",
                                        "down_blocks.0.attentions.1.proj_in")] == [0, 1, 1, 2, 3]
    # element counts at the SD-1.5 widths, from the per-site sizes of SURVEY Appendix A
    C = {"down_blocks.0": 320, "down_blocks.1": 640, "down_blocks.2": 1280, "mid_block": 1280, "up_blocks.1": 1280,
         "up_blocks.2": 640, "up_blocks.3": 320}

    def site_elems(k, r):
        c = C[".".join(k.split(".")[:2]) if not k.startswith("mid_block") else "mid_block"]
        if is_kv(k):
            return (768 + c) * r
        if k.endswith("ff.net.0.proj"):
            return 9 * c * r
        if k.endswith("ff.net.2"):
            return 5 * c * r
        return 2 * c * r
    for r, kv_last, bound in ((32, True, 16 << 20), (320, False, 64 << 20)):
        order, ends = bank_stages(keys, kv_last)
        total = sum(site_elems(k, r) for k in keys)
        assert total == 423936 * r
        late = sum(site_elems(keys[i], r) for i in order[ends[2]:])
        assert 4 * (late + 48 * r) <= bound, (r, 4 * late)


def test_step_input_feed_copies_into_the_static_buffers():
    """ppft._feed: fresh step inputs -> the captured step's static buffers (one multi-tensor copy per dtype); an input that IS the static
    buffer is left alone, dtypes are preserved, the buffers keep their addresses (a replayed graph reads them)."""
    import torch
    from aqualora_amd.ppft import _feed
    static = {"z": torch.zeros(2, 4, 8, 8), "msg": torch.zeros(2, 48), "eps": torch.zeros(2, 4, 8, 8),
              "t": torch.zeros(2, dtype=torch.int64), "ctx": torch.zeros(2, 77, 16, dtype=torch.bfloat16)}
    ptrs = {k: v.data_ptr() for k, v in static.items()}
    g = torch.Generator().manual_seed(0)
    new = {"z": torch.randn(2, 4, 8, 8, generator=g), "msg": torch.randint(0, 2, (2, 48), generator=g).float(),
           "eps": torch.randn(2, 4, 8, 8, generator=g), "t": torch.tensor([17, 933]), "ctx": torch.randn(2, 77, 16, generator=g).to(torch.bfloat16)}
    _feed(static, **new)
    for k in static:
        assert static[k].data_ptr() == ptrs[k] and static[k].dtype == new[k].dtype and torch.equal(static[k], new[k]), k
    before = static["eps"].clone()
    _feed(static, new["z"] * 2, new["msg"], static["eps"], new["t"] + 1, new["ctx"])   # eps IS the static buffer: untouched
    assert torch.equal(static["eps"], before) and torch.equal(static["z"], new["z"] * 2) and static["t"].tolist() == [18, 934]


def test_make_comm_without_a_process_group_is_no_exchange():
    """Single process, no torch.distributed group: no exchange at all (whatever AQL_COMM says)."""
    from aqualora_amd import dp
    assert dp.make_comm(None) == (None, "no exchange (single rank)")
    assert not dp.exchange_active(None)
