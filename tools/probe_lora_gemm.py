"""Parity of the one-launch rank-32 LoRA linear (aql_lora_gemm_fused: 4-wave and wave-specialised kernels, every tile
configuration, ragged M, N not a multiple of 160, bias + residual, per-sample scale rows) against fp32 torch.  Prints
PASS/FAIL lines."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from aqualora_amd import _lib as L  # noqa: E402

dev = "cuda"
torch.manual_seed(0)
ok_all = True
rnd = lambda *s, std=1.0: (torch.randn(*s, device=dev) * std).to(torch.bfloat16)  # noqa: E731


def rel(a, b):
    return float((a.float() - b).abs().max() / b.abs().max())


CASES = [(16384, 320, 320, 4), (16384, 320, 1280, 4), (4096, 640, 640, 4), (4096, 640, 2560, 4), (1024, 1280, 1280, 4),
         (1024, 10240, 1280, 4), (4096, 5120, 640, 2), (308, 640, 768, 4), (1000, 192, 256, 5), (77, 320, 768, 1), (640, 64, 64, 2),
         (2048, 1280, 5120, 2), (300, 320, 2560, 3)]
for M, N, K, nb in CASES:
    rps = (M + nb - 1) // nb
    X, W, A, Bup, S = rnd(M, K), rnd(N, K, std=K ** -0.5), rnd(32, K, std=K ** -0.5), rnd(N, 32, std=0.2), rnd(nb, 32)
    bias, R = rnd(N, std=0.1), rnd(M, N)
    Y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    T = torch.empty(M, 32, dtype=torch.bfloat16, device=dev)
    Ts = torch.empty_like(T)
    rc = L.call_raw("aql_lora_gemm_fused", L.ptr(X), K, L.ptr(W), K, M, N, K, L.ptr(A), L.ptr(S), rps, L.ptr(Bup), L.ptr(bias),
                    L.ptr(R), N, L.ptr(Y), N, L.ptr(T), L.ptr(Ts), 0, L.stream_ptr())
    if rc == 100:
        print(f"PASS lora_gemm M{M} N{N} K{K}: routed to the two-launch path (rc=100)")
        continue
    L.check(rc, "aql_lora_gemm_fused")
    Tr = X.float() @ A.float().t()
    rows = torch.arange(M, device=dev) // rps
    Tb = Tr.to(torch.bfloat16).float()
    Tsr = (Tb * S.float()[rows]).to(torch.bfloat16).float()
    Yr = X.float() @ W.float().t() + Tsr @ Bup.float().t() + bias.float() + R.float()
    eT, eTs, eY = rel(T, Tr), rel(Ts, Tsr), rel(Y, Yr)
    ok = eT < 1e-2 and eTs < 1.5e-2 and eY < 1.5e-2
    ok_all &= ok
    print(f"{'PASS' if ok else 'FAIL'} lora_gemm M{M} N{N} K{K} nb{nb}: T {eT:.2e} Ts {eTs:.2e} Y {eY:.2e}")

# ---- grouped launch (aql_lora_gemm_fused_grouped): G linears that share X == G separate launches, bit for bit
import ctypes  # noqa: E402
for M, K, widths, nb in [(8192, 320, (320, 320, 320), 4), (616, 768, (320, 320, 640, 640, 1280, 1280, 1280), 8),
                         (2048, 1280, (1280, 1280, 1280), 2), (300, 640, (640, 160), 3)]:
    rps = (M + nb - 1) // nb
    G = len(widths)
    N = sum(widths)
    cols = [0]
    for w in widths:
        cols.append(cols[-1] + w)
    X, W, A, Bup, S = rnd(M, K), rnd(N, K, std=K ** -0.5), rnd(32 * G, K, std=K ** -0.5), rnd(N, 32, std=0.2), rnd(nb, 32)
    Y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    T = torch.empty(G, M, 32, dtype=torch.bfloat16, device=dev)
    Ts = torch.empty_like(T)
    cs = (ctypes.c_int * (G + 1))(*cols)
    rc = L.call_raw("aql_lora_gemm_fused_grouped", L.ptr(X), K, L.ptr(W), K, M, N, K, G, cs, L.ptr(A), L.ptr(S), rps, L.ptr(Bup),
                    None, L.ptr(Y), N, L.ptr(T), L.ptr(Ts), 0, L.stream_ptr())
    L.check(rc, "aql_lora_gemm_fused_grouped")
    same, worst = True, 0.0
    for g in range(G):
        c0, c1 = cols[g], cols[g + 1]
        Wg, Ag, Bg = W[c0:c1].contiguous(), A[32 * g:32 * g + 32].contiguous(), Bup[c0:c1].contiguous()
        Yg = torch.empty(M, c1 - c0, dtype=torch.bfloat16, device=dev)
        Tg = torch.empty(M, 32, dtype=torch.bfloat16, device=dev)
        Tsg = torch.empty_like(Tg)
        rc = L.call_raw("aql_lora_gemm_fused", L.ptr(X), K, L.ptr(Wg), K, M, c1 - c0, K, L.ptr(Ag), L.ptr(S), rps, L.ptr(Bg), None,
                        None, 0, L.ptr(Yg), c1 - c0, L.ptr(Tg), L.ptr(Tsg), 0, L.stream_ptr())
        if rc == 0:
            same &= torch.equal(Yg, Y[:, c0:c1]) and torch.equal(Tg, T[g]) and torch.equal(Tsg, Ts[g])
        rows = torch.arange(M, device=dev) // rps
        Tb = (X.float() @ Ag.float().t()).to(torch.bfloat16).float()
        Yr = X.float() @ Wg.float().t() + (Tb * S.float()[rows]).to(torch.bfloat16).float() @ Bg.float().t()
        worst = max(worst, rel(Y[:, c0:c1], Yr))
    ok = same and worst < 1.5e-2
    ok_all &= ok
    print(f"{'PASS' if ok else 'FAIL'} lora_gemm_grouped M{M} K{K} groups{widths}: equals the separate launches {same}, vs fp32 {worst:.2e}")
# K-grouped form (aql_lora_gemm_fused_kgroups): up to 3 LoRA linears summed into one output (q | k | v backward-data)
for (M, N, Ks, nb, res) in [(16384, 320, (320, 320, 320), 4, False), (4096, 640, (640, 640, 640), 4, True),
                            (1000, 1280, (1280, 1280, 1280), 2, False), (4000, 640, (640, 320), 4, False), (512, 320, (320, 320, 320), 2, False)]:
    G = len(Ks)
    rps = (M + nb - 1) // nb
    Xs = [rnd(M, k) for k in Ks]
    Ws = [rnd(N, k, std=k ** -0.5) for k in Ks]
    As = [rnd(32, k, std=k ** -0.5) for k in Ks]
    Bs = [rnd(N, 32, std=0.2) for _ in Ks]
    S = rnd(nb, 32)
    R = rnd(M, N) if res else None
    Y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    T = torch.empty(G, M, 32, dtype=torch.bfloat16, device=dev)
    Ts = torch.empty_like(T)
    vp, lp_, ip = ctypes.c_void_p * G, ctypes.c_long * G, ctypes.c_int * G
    rc = L.call_raw("aql_lora_gemm_fused_kgroups", G, vp(*[t.data_ptr() for t in Xs]), lp_(*Ks), vp(*[t.data_ptr() for t in Ws]), lp_(*Ks),
                    ip(*Ks), vp(*[t.data_ptr() for t in As]), vp(*[t.data_ptr() for t in Bs]), M, N, L.ptr(S), rps, L.ptr(R),
                    0 if R is None else N, L.ptr(Y), N, vp(*[T[g].data_ptr() for g in range(G)]), vp(*[Ts[g].data_ptr() for g in range(G)]),
                    L.stream_ptr())
    if rc == 100:
        print(f"PASS lora_gemm_kgroups M{M} N{N} K{Ks}: no one-round tile, routed to the chained launches (rc=100)")
        continue
    L.check(rc, "aql_lora_gemm_fused_kgroups")
    rows = torch.arange(M, device=dev) // rps
    Yr = torch.zeros(M, N, device=dev)
    worst_t = 0.0
    for g in range(G):
        Tf = Xs[g].float() @ As[g].float().t()
        Tb = Tf.to(torch.bfloat16).float()
        Tsb = (Tb * S.float()[rows]).to(torch.bfloat16).float()
        Yr += Xs[g].float() @ Ws[g].float().t() + Tsb @ Bs[g].float().t()
        worst_t = max(worst_t, rel(T[g], Tf), rel(Ts[g], Tsb))
    if R is not None:
        Yr += R.float()
    e = rel(Y, Yr)
    ok = e < 1.5e-2 and worst_t < 1.5e-2
    ok_all &= ok
    print(f"{'PASS' if ok else 'FAIL'} lora_gemm_kgroups M{M} N{N} K{Ks} res={res}: Y vs fp32 {e:.2e}, T / Ts {worst_t:.2e}")
print("ALL PASS" if ok_all else "SOME FAILED")
