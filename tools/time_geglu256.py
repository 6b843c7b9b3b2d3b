"""HIP-graph timing of ff.net.0.proj + rank-32 LoRA + GEGLU (aql_lora_gemm_fused_geglu) at the three U-Net levels of the twin batch,
default tile picker against the 256 x 256 persistent kernel (AQL_LORA_CFG=t256, aql_gemm_lora_t256.cuh), 4 rotating operand sets.
usage: python tools/time_geglu256.py [B]      (B = samples per half of the twin batch, default 4)"""
import os, sys
os.environ["AQL_LORA_T256"] = "0"     # the default column is the 128 x 160 picker; t256 is forced through AQL_LORA_CFG
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aqualora_amd import _lib as L
dev = "cuda"
torch.manual_seed(0)
rnd = lambda *s: (torch.randn(*s, device=dev)).to(torch.bfloat16)   # noqa: E731
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
NSET = 4


def bench(HW, K, F, twin=True):
    M = (2 if twin else 1) * B * HW
    sets = []
    for _ in range(NSET):
        X, W, bias, A, Bu = rnd(M, K), rnd(2 * F, K) * K ** -0.5, rnd(2 * F) * 0.02, rnd(32, K) / 32, rnd(2 * F, 32) * 0.02
        S = torch.randn(M // HW, 32, device=dev).to(torch.bfloat16)
        if twin:
            S[:B] = 0
        H = torch.empty(M, 2 * F, dtype=torch.bfloat16, device=dev)
        G = torch.empty(M, F, dtype=torch.bfloat16, device=dev)
        T = torch.empty(M, 32, dtype=torch.bfloat16, device=dev)
        sets.append((X, W, bias, A, Bu, S, H, G, T, torch.empty_like(T)))

    def call(i):
        X, W, bias, A, Bu, S, H, G, T, Ts = sets[i]
        rc = L.call_raw("aql_lora_gemm_fused_geglu", L.ptr(X), K, L.ptr(W), K, M, F, K, L.ptr(A), L.ptr(S), HW, L.ptr(Bu), L.ptr(bias),
                        L.ptr(H), 2 * F, L.ptr(G), F, L.ptr(T), L.ptr(Ts), M // 2 if twin else 0, L.stream_ptr())
        assert rc == 0

    def gtime(iters=16):
        call(0)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(iters):
                call(i % NSET)
        g.replay()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.replay()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / iters)
        return best
    res = {}
    for cfg in ("", "t256"):
        if cfg:
            os.environ["AQL_LORA_CFG"] = cfg
        else:
            os.environ.pop("AQL_LORA_CFG", None)
        res[cfg] = gtime()
    os.environ.pop("AQL_LORA_CFG", None)
    fl = 2.0 * M * 2 * F * K + (2.0 * M * 32 * (K + 2 * F)) * (0.5 if twin else 1.0)
    print(f"M {M:6d} K {K:5d} F {F:5d} {'twin' if twin else 'bwd '}: default {res['']:7.1f} us   t256 {res['t256']:7.1f} us  ({res['t256'] / res['']:.3f})"
          f"   t256 = {fl / res['t256'] / 1e6:.0f} TFLOP/s", flush=True)


for HW, K, F in ((4096, 320, 1280), (1024, 640, 2560), (256, 1280, 5120), (64, 1280, 5120)):
    bench(HW, K, F, True)
