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
    q.put((rank, torch.allclose(a, want, atol=1e-6), torch.equal(a, b), not torch.equal(gathered[0], gathered[1])))
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
