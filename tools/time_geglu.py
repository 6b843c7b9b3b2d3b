"""HIP-graph timing of the heaviest one-launch LoRA linear (ff.net.0.proj + LoRA + GEGLU, twin batch at the 64x64 level) and of
the ff.net.2 launch that consumes its output, back to back (what the step runs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aqualora_amd import _lib as L
dev = "cuda"
torch.manual_seed(0)
rnd = lambda *s: (torch.randn(*s, device=dev)).to(torch.bfloat16)
B = 4
M, K, F = 2 * B * 4096, 320, 1280
NSET = 4
sets = []
for _ in range(NSET):
    X, W, bias, A, Bu = rnd(M, K), rnd(2 * F, K) * K ** -0.5, rnd(2 * F) * 0.02, rnd(32, K) / 32, rnd(2 * F, 32) * 0.02
    W2, A2, Bu2, b2, res = rnd(K, F) * F ** -0.5, rnd(32, F) / 32, rnd(K, 32) * 0.02, rnd(K) * 0.02, rnd(M, K)
    S = torch.cat([torch.zeros(B, 32, device=dev), torch.randn(B, 32, device=dev)]).to(torch.bfloat16)
    H = torch.empty(M, 2 * F, dtype=torch.bfloat16, device=dev); G = torch.empty(M, F, dtype=torch.bfloat16, device=dev)
    T = torch.empty(M, 32, dtype=torch.bfloat16, device=dev); Ts = torch.empty_like(T); T2 = torch.empty_like(T); Ts2 = torch.empty_like(T)
    Y = torch.empty(M, K, dtype=torch.bfloat16, device=dev)
    sets.append((X, W, bias, A, Bu, S, H, G, T, Ts, W2, A2, Bu2, b2, res, T2, Ts2, Y))
def geglu(i):
    X, W, bias, A, Bu, S, H, G, T, Ts = sets[i][:10]
    rc = L.call_raw("aql_lora_gemm_fused_geglu", L.ptr(X), K, L.ptr(W), K, M, F, K, L.ptr(A), L.ptr(S), 4096, L.ptr(Bu), L.ptr(bias),
                    L.ptr(H), 2 * F, L.ptr(G), F, L.ptr(T), L.ptr(Ts), M // 2, L.stream_ptr())
    assert rc == 0
def ff2(i):
    X, W, bias, A, Bu, S, H, G, T, Ts, W2, A2, Bu2, b2, res, T2, Ts2, Y = sets[i]
    rc = L.call_raw("aql_lora_gemm_fused", L.ptr(G), F, L.ptr(W2), F, M, K, F, L.ptr(A2), L.ptr(S), 4096, L.ptr(Bu2), L.ptr(b2),
                    L.ptr(res), K, L.ptr(Y), K, L.ptr(T2), L.ptr(Ts2), M // 2, L.stream_ptr())
    assert rc == 0
def gtime(fn, iters=16):
    fn(0); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(iters): fn(i % NSET)
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / iters)
    return best
a, b = gtime(geglu), gtime(ff2)
c = gtime(lambda i: (geglu(i), ff2(i)))
print(f"geglu launch {a:.1f} us   ff.net.2 launch {b:.1f} us   both back to back {c:.1f} us", flush=True)
