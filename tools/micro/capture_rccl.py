import os, torch, torch.distributed as dist
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
x = torch.ones(1 << 20, device="cuda")
y = torch.zeros_like(x)
dist.all_reduce(x)  # warm-up: communicator init outside capture
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
with torch.cuda.graph(g, capture_error_mode="thread_local"):
    y.copy_(x * 2)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        dist.all_reduce(y, op=dist.ReduceOp.AVG)
    z = x + 1          # overlaps with the collective
    torch.cuda.current_stream().wait_stream(side)
    w = y + z
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
print("captured collective ok", float(w[0]), float(y[0]))
dist.destroy_process_group()
