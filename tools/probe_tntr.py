"""TN transpose-read GEMM (aql_gemm_tn_tr_f32) vs fp32 torch and vs the transpose + NT path, on the rank-320 / batch-8
weight-gradient shapes plus ragged ones.  Prints PASS/FAIL lines and timings."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from aqualora_amd import _lib as L, ops  # noqa: E402

dev = "cuda"
ok_all = True


def timeit(fn, n=20):
    fn()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(n):
        fn()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / n * 1e3


def tntr(U, V, C, alpha=1.0):
    L.call("aql_gemm_tn_tr_f32", L.ptr(U), U.stride(0), L.ptr(V), V.stride(0), U.shape[0], U.shape[1], V.shape[1], alpha,
           L.ptr(C), C.stride(0), L.stream_ptr())


torch.manual_seed(0)
for (M, P, Q) in [(32768, 320, 320), (32768, 2560, 320), (32768, 320, 1280), (8192, 640, 320), (8192, 320, 2560), (2048, 1280, 320),
                  (2048, 320, 5120), (616, 320, 768), (512, 320, 1280), (1000, 40, 72), (77, 320, 768), (130, 8, 136),
                  (16384, 320, 32), (16384, 32, 320), (16384, 2560, 32), (16384, 32, 1280), (4096, 640, 32), (4096, 32, 2560),
                  (1024, 32, 1280), (1024, 5120, 32), (308, 32, 768), (256, 32, 32), (1000, 16, 72), (1000, 72, 8)]:
    U = torch.randn(M, P, device=dev).bfloat16()
    V = torch.randn(M, Q, device=dev).bfloat16()
    C = torch.full((P, Q), 0.5, device=dev)
    tntr(U, V, C, 0.25)
    ref = 0.5 + 0.25 * (U.float().t() @ V.float())
    err = float((C - ref).abs().max() / ref.abs().max())
    ok = err < 2e-5 * max(1.0, (M / 512) ** 0.5)   # fp32 accumulate, split order differs
    ok_all &= ok
    print(f"{'PASS' if ok else 'FAIL'} tn_tr M{M} P{P} Q{Q}: relerr={err:.2e}")
    if M >= 512 and max(P, Q) >= 320:
        Z = torch.zeros(P, Q, device=dev)
        t_new = timeit(lambda: tntr(U, V, Z))
        ops.TN_OLD = True
        t_old = timeit(lambda: ops.gemm_tn_acc(U, V, Z))
        ops.TN_OLD = False
        print(f"     time: tn_tr {t_new:7.1f} us = {2 * M * P * Q / t_new / 1e6:6.0f} TF/s | transpose+NT path {t_old:7.1f} us")
# strided operands (column slices of a wider matrix, as the U-Net's packed heads are)
W = torch.randn(4096, 1024, device=dev).bfloat16()
U, V = W[:, 64:384], W[:, 512:1024]
C = torch.zeros(320, 512, device=dev)
tntr(U, V, C)
ref = U.float().t() @ V.float()
err = float((C - ref).abs().max() / ref.abs().max())
ok_all &= err < 1e-4
print(f"{'PASS' if err < 1e-4 else 'FAIL'} tn_tr strided: relerr={err:.2e}")
# Round 6: the grouped launch through ops.DeferredDW -- problems with a side of 320 / 960 take the 128 x 160 tiles (table "x",
# aql_gemm_tn_tr160_grouped), with and without the operand swap, strided column views, ragged token counts; timed against the
# 128 x 128 tiles (AQL_TNTR160=0 is read once per process: the 128-wide timing comes from aql_gemm_tn_tr_f32 above instead)
dfr = ops.DeferredDW(torch.device(dev, 0))
cases = [(32768, 2560, 320), (32768, 320, 1280), (8192, 640, 320), (8192, 320, 2560), (2048, 1280, 320), (2048, 320, 5120), (616, 320, 768),
         (616, 1280, 320), (1000, 320, 72), (77, 960, 640), (130, 136, 320), (4096, 320, 320)]
outs = []
for (M, P, Q) in cases:
    U = torch.randn(M, P, device=dev).bfloat16()
    V = torch.randn(M, Q, device=dev).bfloat16()
    C = torch.full((P, Q), 0.5, device=dev)
    dfr.add_tn(U, V, C, 0.25)
    outs.append((M, P, Q, U, V, C))
Wd = torch.randn(4096, 2048, device=dev).bfloat16()
Us, Vs = Wd[:, 64:384], Wd[:, 1024:1664]          # strided views: 320 and 640 columns of a 2048-wide buffer
Cs = torch.zeros(320, 640, device=dev)
dfr.add_tn(Us, Vs, Cs, 1.0)
nx = dfr.n["x"]
dfr.flush_tn()
torch.cuda.synchronize()
print(f"     {nx} of {len(cases) + 1} problems on the 128 x 160 table")
ok_all &= nx >= len(cases)
for (M, P, Q, U, V, C) in outs:
    ref = 0.5 + 0.25 * (U.float().t() @ V.float())
    err = float((C - ref).abs().max() / ref.abs().max())
    ok = err < 2e-5 * max(1.0, (M / 512) ** 0.5)
    ok_all &= ok
    print(f"{'PASS' if ok else 'FAIL'} tn_tr160 (grouped) M{M} P{P} Q{Q}: relerr={err:.2e}")
ref = Us.float().t() @ Vs.float()
err = float((Cs - ref).abs().max() / ref.abs().max())
ok_all &= err < 1e-4
print(f"{'PASS' if err < 1e-4 else 'FAIL'} tn_tr160 strided: relerr={err:.2e}")
for (M, P, Q) in [(32768, 2560, 320), (32768, 320, 1280), (8192, 640, 320), (2048, 320, 5120)]:
    U = torch.randn(M, P, device=dev).bfloat16()
    V = torch.randn(M, Q, device=dev).bfloat16()
    Z = torch.zeros(P, Q, device=dev)

    def grouped():
        d2 = ops.DeferredDW(torch.device(dev, 0), max_sites=8)
        d2.add_tn(U, V, Z)
        d2.flush_tn()
    keep = ops.DeferredDW(torch.device(dev, 0), max_sites=8)

    def grouped_fast():
        keep.reset()
        keep.add_tn(U, V, Z)
        keep.flush_tn()
    t160 = timeit(grouped_fast)
    t128 = timeit(lambda: tntr(U, V, Z))
    print(f"     time M{M} P{P} Q{Q}: 128 x 160 tiles {t160:7.1f} us = {2 * M * P * Q / t160 / 1e6:6.0f} TF/s | 128 x 128 tiles (lone launch) {t128:7.1f} us")
print("ALL PASS" if ok_all else "SOME FAILED")
if os.environ.get("AQL_TN_SPLITS"):
    print("(timings above ran with AQL_TN_SPLITS=%s)" % os.environ["AQL_TN_SPLITS"])
