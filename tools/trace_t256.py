"""Phase timeline of the 256 x 256 persistent LoRA + GEGLU kernel (aql_gemm_lora_t256.cuh) on the roofline shape: library built
with -DAQL_T256_TRACE (tools/build_alt.sh t256trace aql_gemm_lora.hip -DAQL_T256_TRACE=1), AQL_LIB=altlib/t256trace.so."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aqualora_amd import _lib as L
dev = "cuda"
rnd = lambda *s: (torch.randn(*s, device=dev)).to(torch.bfloat16)   # noqa: E731
B, HW, K, F = 4, 4096, 320, 1280
M = 2 * B * HW
X, W, bias, A, Bu = rnd(M, K), rnd(2 * F, K) * K ** -0.5, rnd(2 * F) * 0.02, rnd(32, K) / 32, rnd(2 * F, 32) * 0.02
S = torch.randn(2 * B, 32, device=dev).to(torch.bfloat16)
S[:B] = 0
H = torch.empty(M, 2 * F, dtype=torch.bfloat16, device=dev)
G = torch.empty(M, F, dtype=torch.bfloat16, device=dev)
T = torch.empty(M, 32, dtype=torch.bfloat16, device=dev)
Ts = torch.empty_like(T)
buf = torch.zeros(4 * 96, dtype=torch.int64, device=dev)
os.environ["AQL_TRACE_BUF"] = hex(buf.data_ptr())
os.environ["AQL_LORA_CFG"] = "t256"
for _ in range(3):
    rc = L.call_raw("aql_lora_gemm_fused_geglu", L.ptr(X), K, L.ptr(W), K, M, F, K, L.ptr(A), L.ptr(S), HW, L.ptr(Bu), L.ptr(bias),
                    L.ptr(H), 2 * F, L.ptr(G), F, L.ptr(T), L.ptr(Ts), M // 2, L.stream_ptr())
    assert rc == 0
torch.cuda.synchronize()
h = buf.cpu().view(4, 96)
names = ["start", "K0 there", "K loop", "up+bias", "H staged", "H stores", "GEGLU", "barrier", "G staged", "G stores", "end"]
for w in range(4):
    print(f"workgroup {77 if w >> 1 else 0} wavefront {(w & 1) * 4}: cycles since the previous mark")
    row = h[w].tolist()
    for t in range(8):
        seg = row[t * 11:(t + 1) * 11]
        if len(seg) < 11 or seg[-1] == 0:
            break
        prev = row[t * 11 - 1] if t else seg[0]
        out = []
        for k in range(11):
            out.append(f"{names[k]} +{seg[k] - (prev if k == 0 else seg[k - 1])}")
        print("   " + "  ".join(out) + f"   | tile total {seg[-1] - seg[0]}")
