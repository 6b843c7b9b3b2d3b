import os, sys, subprocess, torch
sys.path.insert(0, os.getcwd())
# child mode: compute LN / GN / mse outputs with the library selected by AQL_LIB and save them
if len(sys.argv) > 1:
    from aqualora_amd import ops
    torch.manual_seed(0)
    outs = {}
    for M, C in ((32768, 320), (8192, 640), (2048, 1280), (77, 768)):
        x = torch.randn(M, C, device="cuda").to(torch.bfloat16).requires_grad_(True)
        g = (torch.randn(C, device="cuda") * 0.5 + 1).to(torch.bfloat16); b = (torch.randn(C, device="cuda") * 0.1).to(torch.bfloat16)
        y = ops.layernorm(x, g, b)
        y.backward(torch.randn_like(y))
        outs[f"ln{M}x{C}"] = y.detach().cpu(); outs[f"lnb{M}x{C}"] = x.grad.cpu()
    for B, C, H in ((8, 320, 64), (4, 640, 32), (8, 1280, 16), (4, 2560, 8)):
        x = (torch.randn(B, C, H, H, device="cuda") * 2).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        g = (torch.randn(C, device="cuda") * 0.5 + 1).to(torch.bfloat16); b = (torch.randn(C, device="cuda") * 0.1).to(torch.bfloat16)
        y = ops.groupnorm_silu(x, g, b, 1e-5, True)
        y.backward(torch.randn_like(y))
        outs[f"gn{B}x{C}x{H}"] = y.detach().cpu(); outs[f"gnb{B}x{C}x{H}"] = x.grad.cpu()
    # timing of LN forward / backward inside a graph
    import time
    x = torch.randn(32768, 320, device="cuda").to(torch.bfloat16)
    g = torch.ones(320, device="cuda", dtype=torch.bfloat16); b = torch.zeros_like(g)
    from aqualora_amd import _lib as L
    y = torch.empty_like(x); st = torch.empty(32768, 2, device="cuda"); dx = torch.empty_like(x)
    def f():
        L.call("aql_layernorm_fwd", L.ptr(x), 32768, 320, L.ptr(g), L.ptr(b), 1e-5, L.ptr(y), L.ptr(st), L.stream_ptr())
        L.call("aql_layernorm_bwd", L.ptr(x), L.ptr(y), 32768, 320, L.ptr(g), L.ptr(st), None, L.ptr(dx), L.stream_ptr())
    f(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(20): f()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); [gr.replay() for _ in range(5)]; e1.record(); torch.cuda.synchronize()
    print(sys.argv[1], "LN fwd+bwd 32768x320: %.1f us per pair" % (e0.elapsed_time(e1) * 1e3 / 100))
    torch.save(outs, sys.argv[1])
    sys.exit(0)
env = dict(os.environ)
env["AQL_LIB"] = os.path.join(os.getcwd(), "altlib/head.so")
subprocess.run([sys.executable, __file__, "/tmp/old.pt"], env=env, check=True)
env.pop("AQL_LIB")
subprocess.run([sys.executable, __file__, "/tmp/new.pt"], env=env, check=True)
a, b = torch.load("/tmp/old.pt"), torch.load("/tmp/new.pt")
bad = [k for k in a if not torch.equal(a[k], b[k])]
print("tensors compared", len(a), "differing", bad)
