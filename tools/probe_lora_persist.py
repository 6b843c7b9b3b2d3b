"""The persistent one-launch LoRA linear (lora_gemm_kernel_p, AQL_LORA_CFG=p128) against the one-shot 4-wave kernel on the same
tile (AQL_LORA_CFG=d128s): BIT-IDENTICAL outputs (Y / G / H / T / Ts) on plain, twin (lora_row0), grouped, GEGLU and GEGLU-backward
forms, ragged rows, fewer tiles than resident workgroups, several rounds; and HIP-graph timings of both.  PASS/FAIL lines."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from aqualora_amd import _lib as L  # noqa: E402

dev = "cuda"
ALT = os.environ.get("PROBE_ALT", "p128")      # p128: persistent kernel; s128: three-workgroups-per-CU single-stage kernel
torch.manual_seed(0)
rnd = lambda *s, std=1.0: (torch.randn(*s, device=dev) * std).to(torch.bfloat16)  # noqa: E731
ok_all = True


def graph_time(fn, n=20):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        g.replay()
        t1.record()
        torch.cuda.synchronize()
        best = min(best, t0.elapsed_time(t1) / n * 1e3)
    return best


def run(kind, M, N, K, nb, row0=0, widths=None, time_it=False):
    """kind: plain | geglu | geglu_noh | gbwd | grouped"""
    global ok_all
    rps = (M + nb - 1) // nb
    G = len(widths) if widths else 1
    X, W, A, Bup, S = rnd(M, K), rnd(N, K, std=K ** -0.5), rnd(32 * G, K, std=K ** -0.5), rnd(N, 32, std=0.2), rnd(nb, 32)
    if row0:
        S[: row0 // rps] = 0
    bias, R = rnd(N, std=0.1), rnd(M, N)
    F = N // 2
    Hs = rnd(M, 2 * N)   # saved pre-activation of the GEGLU-backward form
    outs = {}

    def call(cfg, keep):
        os.environ["AQL_LORA_CFG"] = cfg
        Y = torch.full((M, 2 * N if kind == "gbwd" else N), float("nan"), dtype=torch.bfloat16, device=dev)
        Gt = torch.full((M, F), float("nan"), dtype=torch.bfloat16, device=dev)
        T = torch.full((G, M, 32), float("nan"), dtype=torch.bfloat16, device=dev)
        Ts = T.clone()
        st = L.stream_ptr()
        if kind == "plain":
            rc = L.call_raw("aql_lora_gemm_fused", L.ptr(X), K, L.ptr(W), K, M, N, K, L.ptr(A), L.ptr(S), rps, L.ptr(Bup), L.ptr(bias),
                            L.ptr(R), N, L.ptr(Y), N, L.ptr(T), L.ptr(Ts), row0, st)
        elif kind in ("geglu", "geglu_noh"):
            rc = L.call_raw("aql_lora_gemm_fused_geglu", L.ptr(X), K, L.ptr(W), K, M, F, K, L.ptr(A), L.ptr(S), rps, L.ptr(Bup),
                            L.ptr(bias), L.ptr(Y) if kind == "geglu" else None, N, L.ptr(Gt), F, L.ptr(T), L.ptr(Ts), row0, st)
        elif kind == "gbwd":
            rc = L.call_raw("aql_lora_gemm_fused_geglu_bwd", L.ptr(X), K, L.ptr(W), K, M, N, K, L.ptr(A), L.ptr(S), rps, L.ptr(Bup),
                            L.ptr(Hs), 2 * N, L.ptr(Y), 2 * N, L.ptr(T), L.ptr(Ts), st)
        else:
            cols = [0]
            for w in widths:
                cols.append(cols[-1] + w)
            carr = (ctypes.c_int * len(cols))(*cols)
            rc = L.call_raw("aql_lora_gemm_fused_grouped", L.ptr(X), K, L.ptr(W), K, M, N, K, G, carr, L.ptr(A), L.ptr(S), rps,
                            L.ptr(Bup), L.ptr(bias), L.ptr(Y), N, L.ptr(T), L.ptr(Ts), row0, st)
        assert rc == 0, (kind, cfg, rc, L.load().aql_last_error())
        if keep:
            outs[cfg] = (Y, Gt, T, Ts)

    call("d128s", True)
    call(ALT, True)
    a, b = outs["d128s"], outs[ALT]
    # H rows below row0 of a twin GEGLU are never written by either kernel: compare bit patterns (NaN == NaN).  T / Ts are specified
    # from row0 on only (tiles that straddle row0 write the rows below it, tiles entirely below it do not: depends on the tile height)
    bits = lambda z: z.view(torch.int16)   # noqa: E731
    same = bool(torch.equal(bits(a[0]), bits(b[0])) and torch.equal(bits(a[1]), bits(b[1])) and
                torch.equal(bits(a[2][:, row0:]), bits(b[2][:, row0:])) and torch.equal(bits(a[3][:, row0:]), bits(b[3][:, row0:])))
    fin = torch.isfinite(b[1].float()).all() if kind.startswith("geglu") else torch.isfinite(b[0].float()).all() or row0 > 0
    msg = ""
    if time_it:
        t1 = graph_time(lambda: call("d128s", False))
        t2 = graph_time(lambda: call(ALT, False))
        msg = f"  one-shot {t1:.1f} us  {ALT} {t2:.1f} us  ({t2 / t1:.3f})"
    good = bool(same) and bool(fin)
    ok_all &= good
    tiles = ((M + 127) // 128) * (N // 160)
    print(f"{'PASS' if good else 'FAIL'} {kind:9s} M{M} N{N} K{K} nb{nb} row0 {row0} tiles {tiles}: identical {bool(same)}{msg}", flush=True)
    os.environ.pop("AQL_LORA_CFG", None)


timing = len(sys.argv) > 1 and sys.argv[1] == "time"
# parity: one tile, fewer tiles than workgroups, exactly one round, partial last round, many rounds; ragged rows; twin; K tails
for kind, M, N, K, nb, row0, widths in [
        ("plain", 100, 160, 64, 1, 0, None), ("plain", 1000, 320, 320, 5, 0, None), ("plain", 16384, 320, 320, 4, 0, None),
        ("plain", 16384, 1280, 320, 4, 0, None), ("plain", 8300, 640, 200, 4, 0, None), ("plain", 32768, 320, 1280, 8, 16384, None),
        ("plain", 8192, 640, 640, 8, 4096, None), ("plain", 4000, 960, 328, 4, 2000, None),
        ("geglu", 32768, 2560, 320, 8, 16384, None), ("geglu", 8192, 5120, 640, 8, 4096, None), ("geglu", 2048, 10240, 1280, 8, 1024, None),
        ("geglu_noh", 16384, 2560, 320, 4, 0, None), ("geglu", 3000, 2560, 320, 3, 1000, None),
        ("gbwd", 16384, 1280, 320, 4, 0, None), ("gbwd", 4096, 2560, 640, 4, 0, None),
        ("grouped", 32768, 960, 320, 8, 16384, (320, 320, 320)), ("grouped", 616, 5120, 768, 8, 308, (320, 320, 640, 640, 1280, 1280, 640)),
        ("grouped", 8192, 1920, 640, 8, 0, (640, 640, 640))]:
    run(kind, M, N, K, nb, row0, widths, time_it=timing)
if ALT == "t256":
    # the 256 x 256 GEGLU tile only: one tile, fewer rows than a tile, ragged rows, K tails (not a multiple of 64), row0 inside a tile,
    # the smallest feature count, rows_per_sample below a tile (9 samples under one tile), no bias-free / H-free mixes, many rounds
    # (the 128 x 160 kernel splits value | gate at 80-column tiles, this one at 128: F must be a multiple of 640 for both to run)
    for kind, M, N, K, nb, row0 in [("geglu", 100, 1280, 64, 1, 0), ("geglu", 256, 1280, 320, 2, 128), ("geglu", 1000, 1280, 200, 5, 400),
                                    ("geglu_noh", 777, 2560, 328, 7, 333), ("geglu", 2048, 2560, 72, 64, 1024), ("geglu", 5000, 3840, 640, 5, 0),
                                    ("geglu", 65536, 2560, 320, 16, 32768), ("geglu_noh", 4096, 10240, 1280, 4, 0)]:
        run(kind, M, N, K, nb, row0, None, time_it=timing)
print("ALL PASS" if ok_all else "SOME FAILED")
sys.exit(0 if ok_all else 1)
