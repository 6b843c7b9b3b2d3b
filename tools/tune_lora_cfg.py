"""Tile sweep of the one-launch LoRA linear (aql_lora_gemm_fused / _geglu) on the shapes of the TWIN forward (2B samples)
and of the backward-data pass (B samples): every kernel configuration is forced through AQL_LORA_CFG and timed inside a
HIP graph over rotating operand sets (cold L2).  Prints one line per shape with all timings and the winner.
COLD=1: the regime of the train step -- every launch of the graph reads its OWN weight / A / B panels and a 600 MB fill between
replays evicts them from the L2s and the Infinity Cache, so weights come from HBM while the activations rotate over NSET sets."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from aqualora_amd import _lib as L  # noqa: E402

dev = "cuda"
NSET = int(os.environ.get("NSET", "6"))
B = int(os.environ.get("B", "4"))
COLD = os.environ.get("COLD", "0") == "1"
FLUSH = torch.empty(600 << 20, dtype=torch.uint8, device="cuda") if COLD else None
CFGS = os.environ.get("CFGS", "auto,w128,w64,w32,d128,d128s,d64,d64s,d32,d32s").split(",")
rnd = lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16)  # noqa: E731


def graph_time(fns, n=24):
    for f in fns:
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(n):
            fns[i % len(fns)]()
    g.replay()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if COLD:
        ts = []
        for _ in range(3):
            FLUSH.fill_(1)
            t0.record()
            g.replay()
            t1.record()
            torch.cuda.synchronize()
            ts.append(t0.elapsed_time(t1) / n * 1e3)
        return sorted(ts)[1]
    t0.record()
    for _ in range(4):
        g.replay()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / (4 * n) * 1e3


def shapes():
    out = []
    for C, N in ((320, 4096), (640, 1024), (1280, 256)):
        for mult, tag in ((2 * B, "twin-fwd"), (B, "bwd")):
            M = mult * N
            out.append((tag, M, C, C, False))           # attention / 1x1 projections
            out.append((tag, M, C, 4 * C, False))       # ff.net.2 (fwd) / ff.net.0 backward has K = 8C
            if tag == "twin-fwd":
                out.append((tag + "-qkv", M, 3 * C, C, False))    # q | k | v (the grouped launch has this shape)
                out.append((tag + "-geglu", M, 8 * C, C, True))   # ff.net.0 + GEGLU
            else:
                out.append((tag, M, C, 8 * C, False))   # d(ff.net.0): N = C, K = 8C
                out.append((tag, M, 4 * C, C, False))   # d(ff.net.2): N = 4C, K = C
    return out


ONLY = os.environ.get("ONLY")     # e.g. ONLY=qkv: shapes whose tag contains the string
for tag, M, N, K, geglu in shapes():
    if ONLY and ONLY not in tag:
        continue
    def mk():
        X, W, A, Bup, S = rnd(M, K), rnd(N, K), rnd(32, K), rnd(N, 32), rnd(2 * B, 32)
        T, Ts = torch.empty(M, 32, dtype=torch.bfloat16, device=dev), torch.empty(M, 32, dtype=torch.bfloat16, device=dev)
        rps = M // (2 * B)
        if geglu:
            F = N // 2
            G = torch.empty(M, F, dtype=torch.bfloat16, device=dev)
            H = torch.empty(M, N, dtype=torch.bfloat16, device=dev)

            def f():
                return L.call_raw("aql_lora_gemm_fused_geglu", L.ptr(X), K, L.ptr(W), K, M, F, K, L.ptr(A), L.ptr(S), rps, L.ptr(Bup),
                                  None, L.ptr(H), N, L.ptr(G), F, L.ptr(T), L.ptr(Ts), 0, L.stream_ptr())
            return f
        Y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)

        def f():
            return L.call_raw("aql_lora_gemm_fused", L.ptr(X), K, L.ptr(W), K, M, N, K, L.ptr(A), L.ptr(S), rps, L.ptr(Bup), None,
                              None, 0, L.ptr(Y), N, L.ptr(T), L.ptr(Ts), 0, L.stream_ptr())
        return f
    if COLD:   # 24 weight sets (one per launch of the graph), NSET activation sets
        acts = [rnd(M, K) for _ in range(NSET)]
        _rnd = rnd
        cnt = [0]

        def rnd(*shape):   # noqa: F811  (mk() draws X first: hand it a rotating activation set instead)
            if shape == (M, K):
                cnt[0] += 1
                return acts[cnt[0] % NSET]
            return _rnd(*shape)
        fns = [mk() for _ in range(24)]
        rnd = _rnd
    else:
        fns = [mk() for _ in range(NSET)]
    res = {}
    for cfg in CFGS:
        if cfg == "auto":
            os.environ.pop("AQL_LORA_CFG", None)
        else:
            os.environ["AQL_LORA_CFG"] = cfg
        rc = fns[0]()
        if rc != 0:
            res[cfg] = float("nan")
            continue
        res[cfg] = graph_time(fns)
    os.environ.pop("AQL_LORA_CFG", None)
    cand = [(v, k) for k, v in res.items() if v == v and (k != "auto" or len(res) == 1)]
    if not cand:
        print(f"{tag:15s} M{M:6d} N{N:6d} K{K:5d}: not served by the one-launch kernel (two-launch form)", flush=True)
        continue
    best = min(cand)
    fl = 2.0 * M * K * (N + 32) + 2.0 * M * 32 * N
    print(f"{tag:15s} M{M:6d} N{N:6d} K{K:5d}: " + " ".join(f"{k}={v:6.1f}" for k, v in res.items()) +
          f"  | best {best[1]} {best[0]:.1f} us = {fl / best[0] / 1e6:.0f} TF/s (auto {res['auto']:.1f})", flush=True)
