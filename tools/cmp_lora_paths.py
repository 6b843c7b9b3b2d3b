"""Per LoRA-linear shape of the train step (twin forward at 2B samples, backward-data at B): the one-launch kernel (picker's
choice) next to the two-launch form (aql_lora_down + aql_gemm_bf16 with Ts.Bup^T as a second K segment -- the GEMM picker may
split K there) and next to the plain GEMM without any LoRA term (the floor of the shape).  HIP-graph timed over rotating
operand sets; every launch of the graph has its own weights (HBM-cold, as in the step)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from aqualora_amd import _lib as L, ops  # noqa: E402

dev = "cuda"
B = int(os.environ.get("B", "4"))
NL = 12   # launches per graph (each with its own weight set)
rnd = lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16)  # noqa: E731
FLUSH = torch.empty(512 << 20, dtype=torch.uint8, device=dev)


def graph_time(fns):
    for f in fns:
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for f in fns:
            f()
    g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        FLUSH.fill_(1)
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        g.replay()
        t1.record()
        torch.cuda.synchronize()
        ts.append(t0.elapsed_time(t1) / len(fns) * 1e3)
    return sorted(ts)[1]


def shapes():
    out = []
    for C, N in ((320, 4096), (640, 1024), (1280, 256), (1280, 64)):
        for mult, tag in ((2 * B, "fwd"), (B, "bwd")):
            M = mult * N
            out.append((tag, M, C, C))
            out.append((tag, M, C, 4 * C))
            if tag == "bwd":
                out.append((tag, M, C, 8 * C))
                out.append((tag, M, 4 * C, C))
            else:
                out.append((tag, M, 3 * C, C))
    return out


for tag, M, N, K in shapes():
    acts = [rnd(M, K) for _ in range(4)]
    res = [rnd(M, N) for _ in range(2)]

    def mk(i, form):
        X = acts[i % 4]
        W, A, Bup, S = rnd(N, K), rnd(32, K), rnd(N, 32), rnd(2 * B, 32)
        Y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        T, Ts = torch.empty(M, 32, dtype=torch.bfloat16, device=dev), torch.empty(M, 32, dtype=torch.bfloat16, device=dev)
        rps = M // (2 * B)
        R = res[i % 2]
        if form == "fused":
            def f():
                return L.call_raw("aql_lora_gemm_fused", L.ptr(X), K, L.ptr(W), K, M, N, K, L.ptr(A), L.ptr(S), rps, L.ptr(Bup), None,
                                  L.ptr(R), N, L.ptr(Y), N, L.ptr(T), L.ptr(Ts), 0, L.stream_ptr())
        elif form == "two":
            def f():
                ops.lora_down(X, K, M, K, A, 32, S, rps, T, Ts)
                ops.gemm_bf16(X, W, None, Ts, Bup, residual=R, out=Y)
                return 0
        elif form == "down":
            def f():
                ops.lora_down(X, K, M, K, A, 32, S, rps, T, Ts)
                return 0
        else:
            def f():
                ops.gemm_bf16(X, W, None, residual=R, out=Y)
                return 0
        return f
    out = {}
    for form in ("fused", "two", "plain", "down"):
        fns = [mk(i, form) for i in range(NL)]
        if fns[0]() == 100:
            out[form] = float("nan")
            continue
        out[form] = graph_time(fns)
    fl = 2.0 * M * K * N
    print(f"{tag} M{M:6d} N{N:5d} K{K:6d}: fused {out['fused']:7.1f}  two-launch {out['two']:7.1f}  plain GEMM {out['plain']:7.1f}  down alone {out['down']:6.1f} us"
          f"   plain = {fl / out['plain'] / 1e6:5.0f} TF/s", flush=True)
