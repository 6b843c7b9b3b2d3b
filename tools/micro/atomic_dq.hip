// Micro-probe (round 6): what the dQ side of a ONE-PASS attention backward would cost in fp32 atomics.
// Owner = one K/V block of KB keys; for every 64-row Q tile it adds a [64][48] fp32 partial of dQ into HBM/L2 with atomicAdd -- per
// (sample, head): (N / KB) x N x 48 x 4 bytes of atomic traffic, (N/KB) adds onto every dQ element.  N = 4096, 8 heads, 4 samples:
// KB = 128: 0.81 GB, 256: 0.40 GB, 512: 0.20 GB.  The kernel does NOTHING else (no loads, no MFMAs): a lower bound of the added time.
// build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/micro/atomic_dq.hip -o tools/micro/atomic_dq && tools/micro/atomic_dq
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(256) void k(float* dq, int N, int KB, int DP) {
  // grid: (N / KB, heads, samples); 4 wavefronts; wavefront w owns rows 16 w .. 16 w + 15 of each 64-row Q tile, lane = (row, 4 columns)
  const int bh = blockIdx.y + gridDim.y * blockIdx.z;
  float* base = dq + (long)bh * N * DP;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int row = 16 * w + (lane & 15), c0 = (lane >> 4) * 4;
  for (int q0 = 0; q0 < N; q0 += 64) {
    float* p = base + (long)(q0 + row) * DP + c0;
#pragma unroll
    for (int f = 0; f < 3; ++f)        // 3 fragments of 16 columns = 48
#pragma unroll
      for (int e = 0; e < 4; ++e) atomicAdd(p + f * 16 + e, 1.0f);
  }
}
int main() {
  const int N = 4096, H = 8, B = 4, DP = 48;
  float* dq;
  hipMalloc(&dq, (size_t)B * H * N * DP * 4);
  hipMemset(dq, 0, (size_t)B * H * N * DP * 4);
  for (int KB : {64, 128, 256, 512, 1024}) {
    dim3 grid(N / KB, H, B);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<<<grid, 256>>>(dq, N, KB, DP);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) k<<<grid, 256>>>(dq, N, KB, DP);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double gb = (double)(N / KB) * N * DP * 4 * H * B / 1e9;
    printf("KB = %4d keys per owner: %5.2f GB of fp32 atomics per backward (4 samples x 8 heads x 4096 tokens), %7.1f us  (%.2f TB/s, %d workgroups)\n",
           KB, gb, ms / 10 * 1e3, gb / (ms / 10 * 1e-3) / 1e3, (N / KB) * H * B);
  }
  return 0;
}
