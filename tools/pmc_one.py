"""One GEMM / conv shape, a handful of launches: the workload for rocprofv3 --pmc passes (tools/pmc_run.sh).
usage: pmc_one.py gemm M N K | conv B H Cin Cout | lora M N K | geglu M F K | attn B S heads | attnq B S heads | tntr M P Q | chain M"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aqualora_amd import _lib as L
dev = "cuda"
torch.manual_seed(0)
ws = torch.empty(64 << 20, dtype=torch.float32, device=dev)
rnd = lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16)
kind = sys.argv[1]; a = [int(v) for v in sys.argv[2:]]
NSET = 6
if kind == "gemm":
    M, N, K = a
    def mk():
        A = rnd(M, K); B = rnd(N, K); C = torch.empty(M, N, dtype=torch.bfloat16, device=dev); bias = rnd(N)
        return lambda: L.call("aql_gemm_bf16", L.ptr(A), K, L.ptr(B), K, M, N, K, None, 0, None, 0, 0, L.ptr(bias), None, 1,
                              None, 0, L.ptr(C), N, L.ptr(ws), ws.numel() * 4, L.stream_ptr())
elif kind == "lora":   # one-launch LoRA linear, rank 32
    M, N, K = a
    def mk():
        X, W, A, Bup, S = rnd(M, K), rnd(N, K), rnd(32, K), rnd(N, 32), rnd(8, 32)
        Y = torch.empty(M, N, dtype=torch.bfloat16, device=dev); T = torch.empty(M, 32, dtype=torch.bfloat16, device=dev); Ts = torch.empty_like(T)
        return lambda: L.call("aql_lora_gemm_fused", L.ptr(X), K, L.ptr(W), K, M, N, K, L.ptr(A), L.ptr(S), M // 8, L.ptr(Bup), None,
                              None, 0, L.ptr(Y), N, L.ptr(T), L.ptr(Ts), 0, L.stream_ptr())
elif kind == "geglu":   # ff.net.0.proj + rank-32 LoRA + GEGLU on a twin batch: LoRA and the pre-activation H on rows >= M/2 only
    M, F, K = a
    def mk():
        X, W, A, Bup, bias = rnd(M, K), rnd(2 * F, K), rnd(32, K), rnd(2 * F, 32), rnd(2 * F)
        S = torch.cat([torch.zeros(4, 32, device=dev), torch.randn(4, 32, device=dev)]).to(torch.bfloat16)
        H = torch.empty(M, 2 * F, dtype=torch.bfloat16, device=dev); G = torch.empty(M, F, dtype=torch.bfloat16, device=dev)
        T = torch.empty(M, 32, dtype=torch.bfloat16, device=dev); Ts = torch.empty_like(T)
        return lambda: L.call("aql_lora_gemm_fused_geglu", L.ptr(X), K, L.ptr(W), K, M, F, K, L.ptr(A), L.ptr(S), M // 8, L.ptr(Bup),
                              L.ptr(bias), L.ptr(H), 2 * F, L.ptr(G), F, L.ptr(T), L.ptr(Ts), M // 2, L.stream_ptr())
elif kind in ("attn", "attnq"):   # self-attention forward + backward, head dim 40 (attnq: q pre-multiplied by d^-1/2 log2 e, the step's form)
    from aqualora_amd import ops
    Bn, S, heads = a
    def mk():
        q, k, v = (rnd(Bn, S, heads * 40).requires_grad_(True) for _ in range(3))
        do = rnd(Bn, S, heads * 40)
        def f():
            o = ops.attention(q, k, v, heads, q_prescaled=(kind == "attnq"))
            o.backward(do)
        return f
elif kind == "chain":   # the row-resident chain  attn.to_out + residual -> LayerNorm -> to_q  on a twin batch (csrc/aql_chain.hip)
    from aqualora_amd import ops
    M, = a
    C = 320
    def mk():
        X, R = rnd(M, C), rnd(M, C)
        S = torch.cat([torch.zeros(4, 32, device=dev), torch.randn(4, 32, device=dev)]).to(torch.bfloat16)
        lin = lambda bias: dict(W=rnd(C, C), bias=rnd(C) if bias else None, Ad=rnd(32, C), Bup=rnd(C, 32), ldw=C)
        e = lambda *sh: torch.empty(*sh, dtype=torch.bfloat16, device=dev)
        st = [dict(lin(True), T=e(M, 32), Ts=e(M, 32), res=R, ldr=C, out=e(M, C), ldo=C, keep=1, ln=1, gamma=rnd(C), beta=rnd(C), eps=1e-5,
                   stats=torch.empty(M, 2, device=dev), nout=e(M, C), ldn=C, nout_row0=M // 2),
              dict(lin(False), T=e(M, 32), Ts=e(M, 32), out=e(M, C), ldo=C, keep=0)]
        return lambda: ops.chain_fwd(X, C, M, M // 8, M // 2, S, st)
elif kind == "tntr":
    from aqualora_amd import ops
    M, P, Q = a
    def mk():
        U, V = rnd(M, P), rnd(M, Q); C = torch.zeros(P, Q, device=dev)
        return lambda: ops.gemm_tn_acc(U, V, C)
else:
    Bn, H, Cin, Cout = a
    def mk():
        xh = rnd(Bn, H, H, Cin); wk = rnd(Cout, 9 * Cin); b = rnd(Cout); y = torch.empty(Bn, H, H, Cout, dtype=torch.bfloat16, device=dev)
        return lambda: L.call("aql_conv3x3_fwd", L.ptr(xh), Bn, H, H, Cin, L.ptr(wk), L.ptr(b), Cout, 1, 0, None, 0, None, L.ptr(y),
                              L.ptr(ws), ws.numel() * 4, L.stream_ptr())
fns = [mk() for _ in range(NSET)]
for f in fns: f()
torch.cuda.synchronize()
