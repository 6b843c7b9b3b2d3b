// Micro-probe: issue cost (shader cycles per wave64 instruction on one SIMD) of v_mfma_f32_16x16x32_bf16 against the
// half-depth v_mfma_f32_16x16x16_bf16 on gfx950, and of the mixed 32 + 16 sequence that would contract a head size of
// 40 (padded to 48) instead of 64.  16 independent accumulators, one wavefront per SIMD and two.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_k16_rate.hip -o tools/micro/mfma_k16_rate && tools/micro/mfma_k16_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));

// mode 0: 16 x (16x16x32)   1: 16 x (16x16x16)   2: 16 x (16x16x32 then 16x16x16 into the same accumulator)
// mode 3: 16 x 32 first, then 16 x 16 (phase-separated)
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters) {
  f32x4_t acc[16];
  bf16x8_t a, b;
  s16x4_t a4, b4;
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[i][e] = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) a[e] = (__bf16)0.001f, b[e] = (__bf16)0.5f;
#pragma unroll
  for (int e = 0; e < 4; ++e) a4[e] = 0x3a83, b4[e] = 0x3f00;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if constexpr (MODE == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
      } else if constexpr (MODE == 1) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc[i], 0, 0, 0);
      } else if constexpr (MODE == 2) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
          acc[(i + 8) & 15] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc[(i + 8) & 15], 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc[i], 0, 0, 0);
      }
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE>
void run(const char* name, int per_iter, int threads) {
  float* out;
  long long* cyc;
  hipMalloc(&out, 4 * 512 * 256);
  hipMalloc(&cyc, 8);
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, cyc, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  long long c;
  hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double n = (double)iters * 4 * per_iter;   // instructions per wavefront
  printf("%-44s waves/SIMD %d: %6.2f s_memtime ticks / instruction / wave   (%.3f ms wall, %.2f ns / instruction / SIMD)\n", name,
         threads / 256, (double)c / n, ms, ms * 1e6 / (n * (threads / 256)));
  hipFree(out), hipFree(cyc);
}

int main() {
  for (int th : {256, 512}) {
    run<0>("16x16x32_bf16", 16, th);
    run<1>("16x16x16_bf16 (1k)", 16, th);
    run<2>("32 + 16 interleaved (per pair: /2)", 32, th);
    run<3>("16 x 32 then 16 x 16 (per pair: /2)", 32, th);
  }
  return 0;
}
