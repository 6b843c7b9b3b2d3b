"""HIP-graph timing of ff.net.2's backward-data launch with the GEGLU backward in its epilogue (aql_lora_gemm_fused_geglu_bwd) on the
64 x 64 level's shape (16384 rows of the watermarked half, F = 1280, K = 320); AQL_LIB selects an ablation build."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aqualora_amd import _lib as L
dev = "cuda"
torch.manual_seed(0)
rnd = lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16)
for M, F, K in ((16384, 1280, 320), (4096, 2560, 640), (1024, 5120, 1280)):
    sets = []
    for _ in range(6):
        dY, Wt, Bt, At = rnd(M, K), rnd(F, K) * K ** -0.5, rnd(32, K) * 0.05, rnd(F, 32) / 32
        S = torch.randn(4, 32, device=dev).to(torch.bfloat16)
        H = rnd(M, 2 * F); DH = torch.empty(M, 2 * F, dtype=torch.bfloat16, device=dev)
        T = torch.empty(M, 32, dtype=torch.bfloat16, device=dev); Ts = torch.empty_like(T)
        sets.append((dY, Wt, Bt, At, S, H, DH, T, Ts))
    def call(i):
        dY, Wt, Bt, At, S, H, DH, T, Ts = sets[i]
        rc = L.call_raw("aql_lora_gemm_fused_geglu_bwd", L.ptr(dY), K, L.ptr(Wt), K, M, F, K, L.ptr(Bt), L.ptr(S), M // 4, L.ptr(At),
                        L.ptr(H), 2 * F, L.ptr(DH), 2 * F, L.ptr(T), L.ptr(Ts), L.stream_ptr())
        assert rc == 0, rc
    for i in range(6): call(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(24): call(i % 6)
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1000 / 24)
    mb = (M * K * 2 + M * 2 * F * 2 * 2) / 1e6
    print(f"{os.environ.get('AQL_LIB', 'product'):>22}: M {M} F {F} K {K}: {best:6.1f} us  ({mb:.0f} MB -> {mb / best * 1e-3:.2f} TB/s)", flush=True)
