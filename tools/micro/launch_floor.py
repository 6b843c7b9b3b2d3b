import torch, time
x = torch.zeros(1024, device="cuda")
def run(n):
    for _ in range(n): x.add_(1.0)
for mode in ("eager", "graph"):
    n = 2000
    if mode == "graph":
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s): run(10)
        torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g): run(n)
        f = g.replay
    else:
        f = lambda: run(n)
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): f()
    torch.cuda.synchronize()
    print(mode, "us per dependent tiny kernel:", (time.perf_counter() - t0) / 5 / n * 1e6)
