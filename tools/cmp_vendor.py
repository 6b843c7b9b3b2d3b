"""Calibration only (never a product path): the step's GEMM / 3x3 conv / attention shapes through the vendor libraries torch
ships with (hipBLASLt / rocBLAS via F.linear, MIOpen via F.conv2d channels-last, the flash kernel via
F.scaled_dot_product_attention) next to our own launches of the same shapes.  HIP-graph timed over rotating HBM-cold operand
sets like tools/cmp_lora_paths.py.  Answers "is 14-20 us for a 6.7 GFLOP projection a property of the shape on this chip or of
our kernel"."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from aqualora_amd import _lib as L, ops  # noqa: E402

dev = "cuda"
B = int(os.environ.get("B", "4"))
NL = 12
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


def eager_time(fn, iters=20):
    """Vendor SDPA is timed eagerly (its launch path is not capture-safe on this build): back-to-back launches, events around."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(iters):
            fn()
        t1.record()
        torch.cuda.synchronize()
        ts.append(t0.elapsed_time(t1) / iters * 1e3)
    return sorted(ts)[1]


def gemm_shapes():
    out = []
    for C, N in ((320, 4096), (640, 1024), (1280, 256), (1280, 64)):
        for mult, tag in ((2 * B, "fwd"), (B, "bwd")):
            M = mult * N
            out.append((tag, M, C, C))
            out.append((tag, M, C, 4 * C))
            if tag == "fwd":
                out.append((tag, M, 3 * C, C))
                out.append((tag, M, 8 * C, C))
            else:
                out.append((tag, M, 4 * C, C))
    return out


what = os.environ.get("WHAT", "gemm,conv,attn").split(",")

if "gemm" in what:
    print("# plain GEMM Y = X W^T (bf16, fp32 accumulate), us per launch, HBM-cold weights")
    GS = gemm_shapes()
    if os.environ.get("GEMM_SHAPES"):   # "M,N,K;..." -- tuning sweeps (AQL_TILE / AQL_W) on a few shapes
        GS = [("fwd",) + tuple(int(v) for v in t.split(",")) for t in os.environ["GEMM_SHAPES"].split(";")]
    for tag, M, N, K in GS:
        acts = [rnd(M, K) for _ in range(4)]

        def mk(i, form):
            X = acts[i % 4]
            W = rnd(N, K)
            Y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            if form == "ours":
                return lambda: ops.gemm_bf16(X, W, None, out=Y)
            return lambda: torch.mm(X, W.t(), out=Y)
        t_o = graph_time([mk(i, "ours") for i in range(NL)])
        t_v = float("nan") if os.environ.get("NO_VENDOR") else graph_time([mk(i, "vendor") for i in range(NL)])
        fl = 2.0 * M * K * N
        print(f"{tag} M{M:6d} N{N:5d} K{K:6d}: ours {t_o:7.1f} ({fl / t_o / 1e6:5.0f} TF/s)  hipBLASLt {t_v:7.1f} ({fl / t_v / 1e6:5.0f} TF/s)"
              f"  ours/vendor {t_o / t_v:5.2f}", flush=True)

if "conv" in what:
    print("# 3x3 conv, stride 1, pad 1, channels-last bf16; us per launch")
    CONV = ((2 * B, 64, 320, 320), (B, 64, 320, 320), (2 * B, 32, 640, 640), (B, 32, 640, 640), (2 * B, 16, 1280, 1280),
                          (B, 16, 1280, 1280), (2 * B, 8, 1280, 1280), (2 * B, 64, 640, 320), (2 * B, 32, 1280, 640), (2 * B, 16, 2560, 1280))
    if os.environ.get("CONV_SHAPES"):   # "B,H,Ci,Co;..." -- tuning sweeps (AQL_SPLITS / AQL_TILE / AQL_W) on a few shapes
        CONV = tuple(tuple(int(v) for v in t.split(",")) for t in os.environ["CONV_SHAPES"].split(";"))
    for Bn, H, Ci, Co in CONV:
        xs = [rnd(Bn, Ci, H, H).contiguous(memory_format=torch.channels_last) for _ in range(3)]

        _hot = {}

        def mkc(i, form):
            if os.environ.get("HOT") and (form in _hot):
                return _hot[form]
            x = xs[i % 3]
            w = rnd(Co, Ci, 3, 3) * 0.02
            if os.environ.get("HOT") and i > 0:   # every launch reads the SAME weights (L2 / MALL resident): how much of the time is the cold panel?
                return mkc(0, form)
            if form == "ours":
                p = ops.PackedConv3x3(w, torch.zeros(Co, device=dev, dtype=torch.bfloat16), 1)

                def f():
                    with torch.no_grad():
                        return ops.conv3x3(x, p)
                _hot[form] = f
                return f
            wc = w.contiguous(memory_format=torch.channels_last)
            return lambda: F.conv2d(x, wc, None, 1, 1)
        try:
            t_o = graph_time([mkc(i, "ours") for i in range(6)])
        except Exception as e:  # noqa: BLE001
            print("ours failed:", repr(e)[:200])
            t_o = float("nan")
        try:
            t_v = float("nan") if os.environ.get("NO_VENDOR") else graph_time([mkc(i, "vendor") for i in range(6)])
        except Exception as e:  # noqa: BLE001
            print("vendor failed:", repr(e)[:200])
            t_v = float("nan")
        fl = 2.0 * Bn * H * H * Ci * Co * 9
        print(f"B{Bn} {H}x{H} {Ci}->{Co}: ours {t_o:7.1f} ({fl / t_o / 1e6:5.0f} TF/s)  MIOpen {t_v:7.1f} ({fl / t_v / 1e6:5.0f} TF/s)"
              f"  ours/vendor {t_o / t_v:5.2f}", flush=True)

if "attn" in what:
    print("# self-attention forward / backward, us per launch (vendor = F.scaled_dot_product_attention)")
    for Bn, Hh, N, d in ((B, 8, 4096, 40), (2 * B, 8, 4096, 40), (B, 8, 1024, 80), (2 * B, 8, 1024, 80), (2 * B, 8, 256, 160)):
        C = Hh * d
        q, k, v, do = (torch.randn(Bn, N, C, device=dev, dtype=torch.bfloat16) for _ in range(4))
        o = torch.empty_like(q)
        lse = torch.empty(Bn, Hh, N, device=dev)
        delta = torch.empty_like(lse)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        ws = torch.empty(16 << 20, device=dev)
        sc = float(d ** -0.5)
        st = L.stream_ptr
        fwd = lambda: L.call("aql_sdpa_fwd", L.ptr(q), C, L.ptr(k), C, L.ptr(v), C, Bn, Hh, N, N, d, sc, L.ptr(o), C, L.ptr(lse), st())  # noqa: E731
        bwd = lambda: L.call("aql_sdpa_bwd", L.ptr(q), C, L.ptr(k), C, L.ptr(v), C, L.ptr(o), L.ptr(do), C, L.ptr(lse), L.ptr(delta), Bn, Hh, N, N,  # noqa: E731
                             d, sc, L.ptr(dq), L.ptr(dk), L.ptr(dv), L.ptr(ws), ws.numel() * 4, st())
        tf = graph_time([fwd] * 6)
        tb = graph_time([bwd] * 6)
        qh, kh, vh = (t.view(Bn, N, Hh, d).transpose(1, 2) for t in (q, k, v))
        line = f"B={Bn} N={N} d={d}: ours fwd {tf:7.1f} bwd {tb:7.1f}"
        for name, be in [(n, getattr(torch.nn.attention.SDPBackend, n)) for n in os.environ.get("SDPA", "FLASH_ATTENTION,EFFICIENT_ATTENTION").split(",")]:
            try:
                with torch.nn.attention.sdpa_kernel(be):
                    vf = lambda: F.scaled_dot_product_attention(qh, kh, vh)  # noqa: E731
                    tvf = eager_time(vf)
                    qg, kg, vg = (t.detach().clone().requires_grad_(True) for t in (qh, kh, vh))
                    og = F.scaled_dot_product_attention(qg, kg, vg)
                    dog = torch.randn_like(og)
                    vb = lambda: torch.autograd.grad(og, (qg, kg, vg), dog, retain_graph=True)  # noqa: E731
                    tvb = eager_time(vb)
                line += f" | {name} fwd {tvf:7.1f} bwd {tvb:7.1f}"
            except Exception as e:  # noqa: BLE001
                line += f" | {name} failed: {repr(e)[:80]}"
        print(line, flush=True)
