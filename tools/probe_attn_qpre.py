"""Self-attention with q pre-multiplied by d^-1/2 log2(e) (aql_sdpa_fwd_qpre / _bwd_qpre, ops.attention(..., q_prescaled=True)) against an
fp32 softmax attention on exactly the operands the kernels see, next to the default path on the unscaled q; `time`: HIP-graph timings."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from aqualora_amd import ops  # noqa: E402

dev = "cuda"
TIME = len(sys.argv) > 1 and sys.argv[1] == "time"
ok_all = True
LOG2E = 1.4426950408889634


def ref(q, k, v, do, H):   # fp32 attention + gradients, q / k / v [B, N, C] fp32 leaves
    B, N, C = q.shape
    d = C // H
    qh, kh, vh = (t.view(B, N, H, d).permute(0, 2, 1, 3) for t in (q, k, v))
    outs = []
    for b in range(B):
        s = torch.einsum("hqd,hkd->hqk", qh[b], kh[b]) * d ** -0.5
        o = torch.einsum("hqk,hkd->hqd", torch.softmax(s, dim=-1), vh[b])
        o.backward(do.view(B, N, H, d).permute(0, 2, 1, 3)[b])
        outs.append(o.detach())
    return torch.stack(outs).permute(0, 2, 1, 3).reshape(B, N, C)


def graph_time(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1000 / reps)
    return best


for name, B, N, H, d, spike in (("64 x 64 level, 4 samples", 4, 4096, 8, 40, 0.0), ("8 samples (twin forward)", 8, 4096, 8, 40, 0.0),
                               ("one sample (128-row workgroups)", 1, 4096, 8, 40, 0.0), ("late spike x3 (fast path)", 2, 4096, 8, 40, 3.0),
                               ("late spike x20 (overflow -> classic pass)", 2, 4096, 8, 40, 20.0), ("ragged N", 2, 1000, 8, 40, 0.0)):
    torch.manual_seed(3)
    C = H * d
    c = d ** -0.5 * LOG2E
    q32 = torch.randn(B, N, C, device=dev) * 1.5
    k16 = (torch.randn(B, N, C, device=dev) * 1.5).to(torch.bfloat16)
    v16 = torch.randn(B, N, C, device=dev).to(torch.bfloat16)
    if spike:
        k16[:, -1] = (k16[:, :64].float().abs().amax() * spike * torch.sign(torch.randn(C, device=dev))).to(torch.bfloat16)
    do = torch.randn(B, N, C, device=dev).to(torch.bfloat16)
    res = {}
    for mode in ("default", "qpre"):
        qk = (q32 * c).to(torch.bfloat16) if mode == "qpre" else q32.to(torch.bfloat16)      # what the producer rounds
        q_eff = (qk.float() / c) if mode == "qpre" else qk.float()                           # the unscaled q the kernel effectively sees
        qa, ka, va = qk.clone().requires_grad_(True), k16.clone().requires_grad_(True), v16.clone().requires_grad_(True)
        o = ops.attention(qa, ka, va, H, q_prescaled=(mode == "qpre"))
        o.backward(do)
        qf, kf, vf = (t.detach().float().requires_grad_(True) for t in (q_eff, k16, v16))
        of = ref(qf, kf, vf, do.float(), H)
        err = lambda a, b_: ((a.float() - b_).abs().max() / b_.abs().max()).item()   # noqa: E731
        res[mode] = (err(o, of), err(qa.grad, qf.grad), err(ka.grad, kf.grad), err(va.grad, vf.grad))
        if TIME:
            fw = lambda: ops.attention(qk, k16, v16, H, q_prescaled=(mode == "qpre"))   # noqa: E731
            with torch.no_grad():
                res[mode] += (graph_time(fw),)
    e = res["qpre"]
    lim = (8e-3, 2e-2, 2e-2, 2e-2) if spike < 10 else (1.2e-2, 4e-2, 4e-2, 4e-2)
    ok = all(x < l_ for x, l_ in zip(e[:4], lim)) and all(x == x for x in e[:4])
    ok_all &= ok
    line = (f"{'PASS' if ok else 'FAIL'} attention qpre [{name}] B{B} N{N} d{d}: o {e[0]:.2e} dq {e[1]:.2e} dk {e[2]:.2e} dv {e[3]:.2e}"
            f"  | default path: o {res['default'][0]:.2e} dq {res['default'][1]:.2e} dk {res['default'][2]:.2e} dv {res['default'][3]:.2e}")
    if TIME:
        line += f"  | forward {res['default'][4]:.1f} -> {res['qpre'][4]:.1f} us"
    print(line, flush=True)
print("ALL PASS" if ok_all else "SOME FAILED")
