"""aql_gemm_bf16_geglu on the 256 x 256 persistent tile (aql_gemm_lora_t256.cuh, SEG2 kernel: the LoRA term of ANY rank as a second K
segment Ts . Bup^T on rows >= row0, or no LoRA at all) against the 128 x 160 kernels (AQL_LORA_CFG=off keeps the tile out): G and H
BIT-IDENTICAL; rows of Ts below row0 are NaN-poisoned (they must not be read).  `time` as first argument adds HIP-graph timings on
config 3's shapes.  PASS/FAIL lines."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from aqualora_amd import _lib as L  # noqa: E402

dev = "cuda"
torch.manual_seed(0)
rnd = lambda *s, std=1.0: (torch.randn(*s, device=dev) * std).to(torch.bfloat16)  # noqa: E731
ok_all = True


def graph_time(fn, n=10):
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


def run(M, F, K, K2, row0, want_h=True, time_it=False):
    global ok_all
    X, W, bias = rnd(M, K), rnd(2 * F, K, std=K ** -0.5), rnd(2 * F, std=0.1)
    Ts = Bup = None
    if K2:
        Ts, Bup = rnd(M, K2), rnd(2 * F, K2, std=0.05)
        Ts[:row0] = float("nan")
    outs = {}

    def call(cfg, keep):
        os.environ["AQL_LORA_CFG"] = cfg
        H = torch.full((M, 2 * F), float("nan"), dtype=torch.bfloat16, device=dev) if want_h else None
        G = torch.full((M, F), float("nan"), dtype=torch.bfloat16, device=dev)
        rc = L.call_raw("aql_gemm_bf16_geglu", L.ptr(X), K, L.ptr(W), K, M, F, K, L.ptr(Ts), K2, L.ptr(Bup), K2, K2, L.ptr(bias),
                        L.ptr(H), 2 * F, L.ptr(G), F, row0, L.stream_ptr())
        if rc == 100 and cfg == "off":
            return False
        assert rc == 0, (cfg, rc, L.load().aql_last_error())
        if keep:
            outs[cfg] = (G, H)
        return True

    if not call("off", True):
        print(f"SKIP M{M} F{F} K{K} K2 {K2}: the 128 x 160 path does not fuse this shape (no reference)", flush=True)
        os.environ.pop("AQL_LORA_CFG", None)
        return
    call("t256", True)
    bits = lambda z: z.view(torch.int16)   # noqa: E731
    same = torch.equal(bits(outs["off"][0]), bits(outs["t256"][0]))
    if want_h:
        same = same and torch.equal(bits(outs["off"][1]), bits(outs["t256"][1]))
    fin = bool(torch.isfinite(outs["t256"][0].float()).all())
    msg = ""
    if time_it:
        t1, t2 = graph_time(lambda: call("off", False)), graph_time(lambda: call("t256", False))
        fl = 2.0 * M * 2 * F * K + 2.0 * (M - row0) * 2 * F * K2
        msg = f"  128x160 {t1:.1f} us  256x256 {t2:.1f} us  ({t2 / t1:.3f}; {fl / t2 / 1e6:.0f} TFLOP/s)"
    good = bool(same) and fin
    ok_all &= good
    print(f"{'PASS' if good else 'FAIL'} M{M} F{F} K{K} K2 {K2} row0 {row0} H {want_h}: identical {bool(same)} finite {fin}{msg}", flush=True)
    os.environ.pop("AQL_LORA_CFG", None)


timing = len(sys.argv) > 1 and sys.argv[1] == "time"
for M, F, K, K2, row0, wh in [(300, 640, 64, 8, 0, True), (600, 640, 320, 320, 128, True), (1000, 1280, 200, 40, 400, True),
                              (777, 640, 328, 16, 333, False), (5000, 1920, 640, 320, 0, True), (4096, 1280, 320, 0, 0, True),
                              (3000, 1280, 320, 0, 1000, False), (2048, 1280, 72, 104, 1024, True)]:
    run(M, F, K, K2, row0, wh)
# config 3 (rank 320, batch 8 twin) and the rank-8 / LoRA-free forms at the sizes the picker takes the tile
for M, F, K, K2, row0 in [(65536, 1280, 320, 320, 32768), (16384, 2560, 640, 320, 8192), (4096, 5120, 1280, 320, 2048),
                          (32768, 1280, 320, 8, 16384), (32768, 1280, 320, 0, 0)]:
    run(M, F, K, K2, row0, True, time_it=timing)
print("ALL PASS" if ok_all else "SOME FAILED")
sys.exit(0 if ok_all else 1)
