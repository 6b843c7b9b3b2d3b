// Shared device/host helpers for the aqualora_hip C-ABI library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <math.h>

typedef uint16_t bf16_t;  // raw bfloat16 bits; no torch / hip_bf16 types cross the ABI

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;   // MFMA A/B fragment (8 bf16 = 4 VGPR)
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;  // 32x32 accumulator
typedef __attribute__((ext_vector_type(4))) float f32x4_t;    // 16x16 accumulator

#define AQL_OK 0
#define AQL_ERR_ARG 1
#define AQL_ERR_HIP 2

// Thread-local last-error text, returned by aql_last_error().
extern "C" const char* aql_last_error(void);
void aql_set_error(const char* fmt, ...);

#define AQL_CHECK_ARG(cond, ...)                 \
  do {                                           \
    if (!(cond)) {                               \
      aql_set_error(__VA_ARGS__);                \
      return AQL_ERR_ARG;                        \
    }                                            \
  } while (0)

#define AQL_CHECK_LAUNCH(name)                                                   \
  do {                                                                           \
    hipError_t _e = hipGetLastError();                                           \
    if (_e != hipSuccess) {                                                      \
      aql_set_error("%s: launch failed: %s", name, hipGetErrorString(_e));       \
      return AQL_ERR_HIP;                                                        \
    }                                                                            \
  } while (0)

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// fp32 -> bf16, round-to-nearest-even: gfx950's v_cvt_pk_bf16_f32 (one VALU op per PAIR; the compiler selects it
// for __bf16 vector conversions, a hand-rolled bit trick costs ~5 ops per element)
typedef __bf16 aql_bf16x2_t __attribute__((ext_vector_type(2)));
typedef float aql_f32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  const aql_f32x2_t v = {lo, hi};
  const aql_bf16x2_t r = __builtin_convertvector(v, aql_bf16x2_t);
  return *reinterpret_cast<const uint32_t*>(&r);
}

__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return (bf16_t)(pack_bf16x2(f, 0.f) & 0xffffu); }

// erf(x) by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7 -- two orders below fp32-in / bf16-out resolution of every
// caller): one v_rcp, one v_exp and six FMAs instead of libm's branchy erff (~4x the VALU work).  Kept for reference builds
// (-DAQL_ERF_7126); the GEGLU paths use the pair form below.
__device__ __forceinline__ float aql_erf(float x) {
#pragma clang fp contract(off)   // every fused multiply-add below is an explicit fmaf: the same bits in every kernel that inlines this
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.f));
  float p = fmaf(t, 1.061405429f, -1.453152027f);
  p = fmaf(t, p, 1.421413741f);
  p = fmaf(t, p, -0.284496736f);
  p = fmaf(t, p, 0.254829592f);
  const float e = __expf(-ax * ax);
  return copysignf(fmaf(-p * t, e, 1.f), x);
}

// Pair form for the GEGLU epilogues (scripts/lib/original_unet.py:721-729, F.gelu exact form), which are VALU-bound: ~10k cycles of
// activation per 128 x 160 tile next to a 5-step K loop.  A&S 7.1.28: erf(x) = 1 - (1 + a1 x + ... + a6 x^6)^-16, |error| <= 3e-7,
// ONE transcendental (v_rcp) per element instead of two (v_rcp + v_exp), and every other operation -- six FMAs, four squarings -- on
// TWO elements per instruction (v_pk_fma_f32 / v_pk_mul_f32: the two bf16 halves of a 32-bit word).  Per element ~11 issue slots
// instead of ~19.  Large |x|: the 16th power overflows to +inf, its reciprocal is 0, erf = +-1 (no NaN: inf * inf = inf).
// Used by the fused epilogues AND the stand-alone geglu kernels, so the two paths stay bit-identical to each other (packed lanes are
// independent IEEE operations: the pairing does not matter).
typedef float aql_f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ aql_f32x2_t aql_splat2(float v) { return aql_f32x2_t{v, v}; }
__device__ __forceinline__ aql_f32x2_t aql_erf2(aql_f32x2_t x) {
#pragma clang fp contract(off)
  const aql_f32x2_t ax = __builtin_elementwise_abs(x);
  aql_f32x2_t p = __builtin_elementwise_fma(ax, aql_splat2(0.0000430638f), aql_splat2(0.0002765672f));
  p = __builtin_elementwise_fma(ax, p, aql_splat2(0.0001520143f));
  p = __builtin_elementwise_fma(ax, p, aql_splat2(0.0092705272f));
  p = __builtin_elementwise_fma(ax, p, aql_splat2(0.0422820123f));
  p = __builtin_elementwise_fma(ax, p, aql_splat2(0.0705230784f));
  p = __builtin_elementwise_fma(ax, p, aql_splat2(1.f));
  p = p * p;
  p = p * p;
  p = p * p;
  p = p * p;
  const aql_f32x2_t r = aql_f32x2_t{__builtin_amdgcn_rcpf(p.x), __builtin_amdgcn_rcpf(p.y)};
  const aql_f32x2_t e = aql_splat2(1.f) - r;
  return aql_f32x2_t{copysignf(e.x, x.x), copysignf(e.y, x.y)};
}
__device__ __forceinline__ aql_f32x2_t aql_gelu2(aql_f32x2_t g) {
#pragma clang fp contract(off)
#ifdef AQL_ERF_7126
  return aql_f32x2_t{0.5f * g.x * (1.f + aql_erf(g.x * 0.70710678118654752f)), 0.5f * g.y * (1.f + aql_erf(g.y * 0.70710678118654752f))};
#else
  const aql_f32x2_t e = aql_erf2(g * aql_splat2(0.70710678118654752f));
  return (aql_splat2(0.5f) * g) * (aql_splat2(1.f) + e);
#endif
}
__device__ __forceinline__ float aql_gelu(float g) { return aql_gelu2(aql_splat2(g)).x; }
// d(value * gelu(gate)) -> d(value), d(gate) (h = value, g = gate) on a pair.  Shared by aql_geglu_bwd and the GEGLU-backward GEMM
// epilogue, which must agree bit for bit: with -ffp-contract=fast the compiler's choice of which a*b+c to fuse depended on the
// code around the inlined body (one element of a 4096x2560 tile rounded differently after an unrelated epilogue change).
__device__ __forceinline__ void aql_geglu_bwd2(aql_f32x2_t d, aql_f32x2_t h, aql_f32x2_t g, aql_f32x2_t& dh, aql_f32x2_t& dg) {
#pragma clang fp contract(off)
#ifdef AQL_ERF_7126
  const aql_f32x2_t e = aql_f32x2_t{aql_erf(g.x * 0.70710678118654752f), aql_erf(g.y * 0.70710678118654752f)};
#else
  const aql_f32x2_t e = aql_erf2(g * aql_splat2(0.70710678118654752f));
#endif
  const aql_f32x2_t cdf = aql_splat2(0.5f) * (aql_splat2(1.f) + e);
  const aql_f32x2_t q = aql_splat2(-0.5f) * g * g;
  const aql_f32x2_t pdf = aql_splat2(0.3989422804014327f) * aql_f32x2_t{__expf(q.x), __expf(q.y)};
  dh = d * g * cdf;
  dg = d * h * __builtin_elementwise_fma(g, pdf, cdf);
}
__device__ __forceinline__ void aql_geglu_bwd1(float d, float h, float g, float& dh, float& dg) {
  aql_f32x2_t a, b;
  aql_geglu_bwd2(aql_splat2(d), aql_splat2(h), aql_splat2(g), a, b);
  dh = a.x;
  dg = b.x;
}

__device__ __forceinline__ float bf16lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// All-lanes reductions over the 64 lanes of a wavefront WITHOUT the LDS pipe.  `__shfl_xor` compiles to ds_bpermute_b32: six dependent
// LDS round trips (~120 cycles each) per reduction -- 12 of them were a quarter of a LayerNorm wavefront's life (one row per wavefront,
// two dependent reductions).  Same butterfly, same operand pairs, so the same bits (IEEE add / max are commutative): the xor-32 and xor-16
// steps are gfx950's v_permlane32_swap / v_permlane16_swap (both results of swap(v, v) together hold own and partner value in every
// lane), the steps inside a 16-lane row are DPP row rotations -- lane (l + n) % 16 is not lane l ^ n when bit log2(n) of l is set, but it
// differs from l ^ n only in the bits the earlier steps have already merged, so it holds the same partial sum.
template <int CTRL>
__device__ __forceinline__ float aql_dpp_row(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
  auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(b[0]) + __uint_as_float(b[1]);
  v += aql_dpp_row<0x128>(v);   // row_ror:8
  v += aql_dpp_row<0x124>(v);   // row_ror:4
  v += aql_dpp_row<0x122>(v);   // row_ror:2
  v += aql_dpp_row<0x121>(v);   // row_ror:1
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
  auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
  v = fmaxf(v, aql_dpp_row<0x128>(v));
  v = fmaxf(v, aql_dpp_row<0x124>(v));
  v = fmaxf(v, aql_dpp_row<0x122>(v));
  v = fmaxf(v, aql_dpp_row<0x121>(v));
  return v;
}

static inline int aql_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Tuning hooks whose A/B is settled (tile pickers, GroupNorm forms, conv row tiles, ring depths ...): environment variables only in
// -DAQL_EXPERIMENTS builds (tools/build_alt.sh <name> <file.hip> -DAQL_EXPERIMENTS, selected with AQL_LIB=altlib/<name>.so); the product
// library carries the measured optimum as a constant.  Round 6: ~45 runtime toggles -> the handful DESIGN.md
// section 6 lists.
#include <stdlib.h>
#ifdef AQL_EXPERIMENTS
#define AQL_TUNE_INT(name, dflt) (getenv(name) ? atoi(getenv(name)) : (dflt))
#else
#define AQL_TUNE_INT(name, dflt) (dflt)
#endif
