"""Repeatability of the persistent 256 x 256 GEGLU kernel (aql_gemm_lora_t256.cuh): the same launch N times on NaN-poisoned outputs,
every result compared bit for bit with the first (a missed barrier / vmcnt wait or a stage reused too early shows up as a difference),
rank-32 one-launch form and SEG2 form, twin batch.  usage: python tools/stress_t256.py [N]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from aqualora_amd import _lib as L  # noqa: E402

dev = "cuda"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
torch.manual_seed(1)
rnd = lambda *s, std=1.0: (torch.randn(*s, device=dev) * std).to(torch.bfloat16)  # noqa: E731
os.environ["AQL_LORA_CFG"] = "t256"
ok = True
for M, F, K, r2 in ((32768, 1280, 320, 0), (8192, 2560, 640, 0), (16384, 1280, 320, 320), (5000, 1280, 200, 40)):
    X, W, bias, A, Bu = rnd(M, K), rnd(2 * F, K, std=K ** -0.5), rnd(2 * F, std=0.1), rnd(32, K, std=K ** -0.5), rnd(2 * F, 32, std=0.2)
    nb = 8
    rps = (M + nb - 1) // nb
    S = rnd(nb, 32)
    row0 = (M // 2 // 8) * 8
    Ts2, Bu2 = (rnd(M, r2), rnd(2 * F, r2, std=0.05)) if r2 else (None, None)
    first = None
    bad = 0
    for it in range(N):
        H = torch.full((M, 2 * F), float("nan"), dtype=torch.bfloat16, device=dev)
        G = torch.full((M, F), float("nan"), dtype=torch.bfloat16, device=dev)
        T = torch.full((M, 32), float("nan"), dtype=torch.bfloat16, device=dev)
        Ts = T.clone()
        if r2:
            rc = L.call_raw("aql_gemm_bf16_geglu", L.ptr(X), K, L.ptr(W), K, M, F, K, L.ptr(Ts2), r2, L.ptr(Bu2), r2, r2, L.ptr(bias),
                            L.ptr(H), 2 * F, L.ptr(G), F, row0, L.stream_ptr())
        else:
            rc = L.call_raw("aql_lora_gemm_fused_geglu", L.ptr(X), K, L.ptr(W), K, M, F, K, L.ptr(A), L.ptr(S), rps, L.ptr(Bu), L.ptr(bias),
                            L.ptr(H), 2 * F, L.ptr(G), F, L.ptr(T), L.ptr(Ts), row0, L.stream_ptr())
        assert rc == 0, rc
        cur = [t.view(torch.int16) for t in ((G, H[row0:]) if r2 else (G, H[row0:], T[row0:], Ts[row0:]))]
        if first is None:
            first = [c.clone() for c in cur]
            assert torch.isfinite(G.float()).all()
        elif not all(torch.equal(a, b) for a, b in zip(first, cur)):
            bad += 1
    ok &= bad == 0
    print(f"{'PASS' if bad == 0 else 'FAIL'} M{M} F{F} K{K} K2 {r2}: {N} launches, {bad} differ from the first", flush=True)
print("ALL PASS" if ok else "SOME FAILED")
sys.exit(0 if ok else 1)
