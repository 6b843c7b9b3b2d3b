"""Minimal workload for the FETCH_SIZE / WRITE_SIZE pass: the dominant conv (3 launches) and the ff.net.0.proj GEMM."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aqualora_amd import ops
dev = "cuda"
x = torch.randn(4, 320, 64, 64, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
pk = ops.PackedConv3x3(torch.randn(320, 320, 3, 3, device=dev) * 0.02, torch.zeros(320, device=dev), 1)
with torch.no_grad():
    for _ in range(3): ops.conv3x3(x, pk)
A2 = torch.randn(16384, 320, device=dev).bfloat16(); B2 = torch.randn(2560, 320, device=dev).bfloat16()
for _ in range(3): ops.gemm_bf16(A2, B2)
torch.cuda.synchronize()
