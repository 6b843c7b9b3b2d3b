"""Does a latency-bound one-round kernel pay for a cold instruction cache?  Times the one-launch LoRA linear (32768 x 320 x 320,
rank 32; lora_gemm_kernel<128,160,64,80,2>, 44 KB of code) inside HIP graphs of 100 launches on rotating operands:
  A only            -- the same code back to back
  A, C alternating  -- a different large kernel (row-tile conv, attention forward) between any two launches of A
and reports (A,C) - (C only) as A's time in the mix."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from aqualora_amd import ops, synth, _lib as L
dev = torch.device("cuda", 0)
NX = 8
M = 32768
X = [synth.normal(f"x{i}", (M, 320), 1.0, 1, dev).to(torch.bfloat16) for i in range(NX)]
W = [synth.normal(f"w{i}", (320, 320), 0.05, 1, dev).to(torch.bfloat16) for i in range(NX)]
A_ = synth.normal("a", (32, 320), 0.05, 1, dev).to(torch.bfloat16)
B_ = synth.normal("b", (320, 32), 0.05, 1, dev).to(torch.bfloat16)
S = synth.normal("s", (8, 32), 1.0, 1, dev).to(torch.bfloat16)
Y = torch.empty(M, 320, dtype=torch.bfloat16, device=dev)
T = torch.empty(M, 32, dtype=torch.bfloat16, device=dev)
Ts = torch.empty_like(T)
def a_call(i):
    rc = L.call_raw("aql_lora_gemm_fused", L.ptr(X[i % NX]), 320, L.ptr(W[i % NX]), 320, M, 320, 320, L.ptr(A_), L.ptr(S), 4096,
                    L.ptr(B_), None, None, 0, L.ptr(Y), 320, L.ptr(T), L.ptr(Ts), 0, L.stream_ptr())
    assert rc == 0, rc
wc = synth.normal("cw", (320, 320, 3, 3), 0.02, 1, dev)
pk = ops.PackedConv3x3(wc, torch.zeros(320, device=dev), 1)
xc = [synth.normal(f"cx{i}", (2, 320, 64, 64), 1.0, 1, dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) for i in range(NX)]
q = [synth.normal(f"q{i}", (1, 4096, 320), 1.0, 1, dev).to(torch.bfloat16) for i in range(3)]
def c_conv(i): ops.conv3x3(xc[i % NX], pk)
def c_attn(i): ops.attention(q[0], q[1], q[2], 8)
def graph_time(fns, n=100):
    with torch.no_grad():
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(n):
                for f in fns: f(i)
        g.replay(); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / n * 1e3)
    return sorted(ts)[2]
ta = graph_time([a_call])
print(f"A only: {ta:.1f} us")
for name, c in (("row-tile conv (2 samples)", c_conv), ("attention fwd (1 sample)", c_attn)):
    tc = graph_time([c]); tac = graph_time([a_call, c])
    print(f"C = {name}: C only {tc:.1f} us, A+C {tac:.1f} us -> A in the mix {tac - tc:.1f} us (alone {ta:.1f})")
