// Micro-probe: issue cost (cycles per wave64 instruction per SIMD) of the VALU ops in the attention softmax, alone and beside
// MFMAs, with 1 and 2 wavefronts per SIMD.  Uses s_memtime around an unrolled block of independent instructions.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rate.hip -o tools/micro/valu_rate && tools/micro/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

// mode: 0 fma, 1 exp, 2 max3, 3 pk_mul, 4 cvt_pk, 5 mfma only, 6 mfma + 3 fma each, 7 mfma + 1 exp + 1 fma each,
//       8 mfma + 2 exp each, 9: 16 mfma then 48 fma (phase-separated), 10: fma+exp pairs (softmax mix), 11 permlane16_swap
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters) {
  float x[16];
  f32x4_t acc[16];
  bf16x8_t a, b;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    x[i] = -0.001f * (threadIdx.x + i);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[i][e] = 0.f;
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) a[e] = (__bf16)0.001f, b[e] = (__bf16)0.5f;
  const float c = 0.999f;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if constexpr (MODE == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[i]) : "v"(c));
      } else if constexpr (MODE == 1) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
      } else if constexpr (MODE == 2) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(x[(i + 1) & 15]), "v"(c));
      } else if constexpr (MODE == 3) {
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
          f32x2_t v = {x[i], x[i + 1]};
          asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v) : "v"(v));
          asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v) : "v"(v));
          x[i] = v[0], x[i + 1] = v[1];
        }
      } else if constexpr (MODE == 4) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          unsigned r;
          asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(x[i]), "v"(x[(i + 1) & 15]));
          x[i] = __uint_as_float(r);
        }
      } else if constexpr (MODE == 5) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
      } else if constexpr (MODE == 6) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
          asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[i]) : "v"(c));
          asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[(i + 5) & 15]) : "v"(c));
          asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[(i + 9) & 15]) : "v"(c));
        }
      } else if constexpr (MODE == 7) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
          asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
          asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[(i + 5) & 15]) : "v"(c));
        }
      } else if constexpr (MODE == 8) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
          asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
          asm volatile("v_exp_f32 %0, %0" : "+v"(x[(i + 7) & 15]));
        }
      } else if constexpr (MODE == 9) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[i]) : "v"(c));
      } else if constexpr (MODE == 10) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[i]) : "v"(c));
          asm volatile("v_exp_f32 %0, %0" : "+v"(x[(i + 8) & 15]));
        }
      } else if constexpr (MODE == 12) {   // legacy K=16 MFMA
        typedef short s4 __attribute__((ext_vector_type(4)));
        s4 a4 = {1, 2, 3, 4}, b4 = {5, 6, 7, 8};
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc[i], 0, 0, 0);
      } else if constexpr (MODE == 13) {   // K=32 + K=16 alternating (the d = 40 contraction as 32 + 8)
        typedef short s4 __attribute__((ext_vector_type(4)));
        s4 a4 = {1, 2, 3, 4}, b4 = {5, 6, 7, 8};
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
          acc[i + 1] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc[i + 1], 0, 0, 0);
        }
      } else if constexpr (MODE == 14) {   // v_max_f32 (2-operand)
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[i]) : "v"(x[(i + 1) & 15]));
      } else if constexpr (MODE == 15) {   // v_mul_f32
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(c));
      } else if constexpr (MODE == 16) {   // v_pk_fma_f32
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
          f32x2_t v = {x[i], x[i + 1]};
          asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(v) : "v"(v));
          asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(v) : "v"(v));
          x[i] = v[0], x[i + 1] = v[1];
        }
      } else if constexpr (MODE >= 17 && MODE <= 20) {   // 32x32x16 MFMA (32 cycles): alone / + 5 fma / + 2 exp + 2 fma / + 1 exp + 1 fma + 1 max3 + 1 cvt
        typedef float f32x16_t __attribute__((ext_vector_type(16)));
        static_assert(sizeof(f32x16_t) == 64, "");
        f32x16_t* big = reinterpret_cast<f32x16_t*>(acc);   // 4 independent 32x32 accumulators in the 16 f32x4 registers
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          big[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, big[i & 3], 0, 0, 0);
          if constexpr (MODE == 18) {
            asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[i]) : "v"(c));
            asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[(i + 3) & 15]) : "v"(c));
            asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[(i + 6) & 15]) : "v"(c));
            asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[(i + 9) & 15]) : "v"(c));
            asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[(i + 12) & 15]) : "v"(c));
          } else if constexpr (MODE == 19) {
            asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
            asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[(i + 3) & 15]) : "v"(c));
            asm volatile("v_exp_f32 %0, %0" : "+v"(x[(i + 6) & 15]));
            asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[(i + 9) & 15]) : "v"(c));
          } else if constexpr (MODE == 20) {
            unsigned r;
            asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
            asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[(i + 3) & 15]) : "v"(c));
            asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[(i + 6) & 15]) : "v"(x[(i + 7) & 15]), "v"(c));
            asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(x[(i + 9) & 15]), "v"(x[(i + 10) & 15]));
            x[(i + 12) & 15] = __uint_as_float(r);
          }
        }
      } else if constexpr (MODE == 11) {
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
          asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(x[i]), "+v"(x[i + 1]));
          asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x[i]), "+v"(x[i + 1]));
        }
      }
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += x[i] + acc[i][0] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int per_iter_valu, int per_iter_mfma) {
  float* out; long long* cyc;
  const int blocks = 256 * 2;
  hipMalloc(&out, blocks * 512 * 4); hipMalloc(&cyc, blocks * 8);
  const int iters = 40000;
  for (int wpb : {256, 512}) {   // 256 threads = 1 wave / SIMD (one workgroup per CU), 512 = 2 waves / SIMD
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(wpb), 0, 0, out, cyc, iters);   // warm the clocks
    hipEventRecord(a);
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(wpb), 0, 0, out, cyc, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    long long h[4]; hipMemcpy(h, cyc, 32, hipMemcpyDeviceToHost);
    // s_memtime counts at 100 MHz on gfx9; report wall-derived cycles at 2.4 GHz instead
    const double cycles = ms * 1e-3 * 2.4e9 / iters / 4.0;   // per unrolled block (4 per iteration)
    const int waves = wpb / 256;
    const double memt = (double)h[0] / iters / 4.0;
    printf("%-34s waves/SIMD %d: %7.1f cyc (wall at 2.4 GHz; s_memtime %.1f) per block of %2d VALU + %2d MFMA per wave  -> %.2f per wave-instr, %.2f SIMD cycles per instr\n", name, waves,
           cycles, memt,
           per_iter_valu, per_iter_mfma, cycles / (per_iter_valu + per_iter_mfma),
           cycles / ((per_iter_valu + per_iter_mfma) * waves));
  }
}

int main() {
  run<0>("v_fma_f32", 16, 0);
  run<1>("v_exp_f32", 16, 0);
  run<2>("v_max3_f32", 16, 0);
  run<3>("v_pk_mul_f32", 16, 0);
  run<4>("v_cvt_pk_bf16_f32", 16, 0);
  run<11>("v_permlane16/32_swap", 16, 0);
  run<10>("fma + exp pairs", 32, 0);
  run<5>("mfma 16x16x32 bf16", 0, 16);
  run<6>("mfma + 3 fma each", 48, 16);
  run<7>("mfma + exp + fma each", 32, 16);
  run<8>("mfma + 2 exp each", 32, 16);
  run<9>("16 mfma, then 48 fma", 48, 16);
  run<12>("mfma 16x16x16 bf16_1k", 0, 16);
  run<13>("mfma K32 + K16 alternating", 0, 16);
  run<14>("v_max_f32", 16, 0);
  run<15>("v_mul_f32", 16, 0);
  run<16>("v_pk_fma_f32", 16, 0);
  run<17>("mfma 32x32x16 bf16", 0, 16);
  run<18>("mfma 32x32x16 + 5 fma each", 80, 16);
  run<19>("mfma 32x32x16 + 2 exp + 2 fma", 64, 16);
  run<20>("mfma 32x32x16 + exp fma max3 cvt", 64, 16);
  return 0;
}
