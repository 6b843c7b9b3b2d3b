// Micro-probe (round 3): does VALU work hide under a 32x32x16 MFMA when the accumulators live in AGPRs instead of VGPRs?
// Four independent accumulators round-robin, NV VALU instructions (fma / exp mix) after each MFMA, 1 and 2 wavefronts per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_agpr.hip -o tools/micro/mfma_agpr && tools/micro/mfma_agpr
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

template <bool AG, int NV, bool EXP>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  float x[16];
  f32x16_t acc[4];
  bf16x8_t a, b;
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = -0.001f * (threadIdx.x + i);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) a[e] = (__bf16)0.001f, b[e] = (__bf16)0.5f;
  const float c = 0.999f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if constexpr (AG) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i & 3]) : "v"(a), "v"(b));
      else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i & 3]) : "v"(a), "v"(b));
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        if (EXP && (v & 1)) asm volatile("v_exp_f32 %0, %0" : "+v"(x[(i + 3 * v) & 15]));
        else asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[(i + 3 * v) & 15]) : "v"(c));
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += x[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <bool AG, int NV, bool EXP>
void run() {
  float* out;
  hipMalloc(&out, 256 * 512 * 4);
  const int iters = 20000;
  for (int wpb : {256, 512}) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<AG, NV, EXP>), dim3(256), dim3(wpb), 0, 0, out, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<AG, NV, EXP>), dim3(256), dim3(wpb), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const int waves = wpb / 256;
    const double per_mfma = ms * 1e-3 * 2.4e9 / iters / 16.0 / waves;   // SIMD cycles (at 2.4 GHz) per MFMA + its NV VALU ops
    printf("acc in %s, %d %s per MFMA, %d waves/SIMD: %6.1f SIMD-cycles per (MFMA + VALU group)\n", AG ? "AGPR" : "VGPR", NV,
           EXP ? "fma/exp alternating" : "fma", waves, per_mfma);
  }
  hipFree(out);
}

int main() {
  run<false, 0, false>(); run<true, 0, false>();
  run<false, 4, false>(); run<true, 4, false>();
  run<false, 8, false>(); run<true, 8, false>();
  run<false, 12, false>(); run<true, 12, false>();
  run<false, 4, true>(); run<true, 4, true>();
  run<false, 8, true>(); run<true, 8, true>();
  return 0;
}
