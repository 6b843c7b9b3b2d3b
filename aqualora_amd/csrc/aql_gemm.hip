// C-ABI entry points built on the MFMA GEMM core (aql_gemm.cuh).  See include/aqualora_hip.h for the
// contract of every symbol and the reference interface (file:line) it replaces.
#include <type_traits>
#include "aql_gemm.cuh"
#include "aql_conv_row.cuh"
#include <stdarg.h>
#include <stdlib.h>

using namespace aqlgemm;

#define AQL_NOT_FUSED 100
static thread_local char g_err[512] = "";
namespace aqlt256 {   // aql_gemm_lora.hip (aql_gemm_lora_t256.cuh)
int t256_geglu_two_segments(const bf16_t* A, long lda, const bf16_t* B, long ldb, long M, int F, int K, const bf16_t* A2, long lda2,
                            const bf16_t* B2, long ldb2, int K2, const bf16_t* bias, bf16_t* H, long ldh, bf16_t* G, long ldg,
                            long row0, hipStream_t stream);
}

extern "C" const char* aql_last_error(void) { return g_err; }
void aql_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

namespace {

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Sum split-K slabs and apply the bf16 epilogue (bias, per-sample row bias, residual).
__global__ __launch_bounds__(256) void splitk_finalize_kernel(const float* __restrict__ slabs, int splits, long M,
                                                              int N, const bf16_t* __restrict__ bias,
                                                              const bf16_t* __restrict__ rowbias,
                                                              long rowbias_ld, int rows_per_sample,
                                                              const bf16_t* __restrict__ residual, long ldr,
                                                              bf16_t* __restrict__ C, long ldc) {
  const long nchunk = M * (N / 4);
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < nchunk; id += (long)gridDim.x * blockDim.x) {
    const long m = id / (N / 4);
    const int n = (int)(id - m * (N / 4)) * 4;
    float4 s = *reinterpret_cast<const float4*>(slabs + m * N + n);
    for (int z = 1; z < splits; ++z) {
      const float4 t = *reinterpret_cast<const float4*>(slabs + ((long)z * M + m) * N + n);
      s.x += t.x;
      s.y += t.y;
      s.z += t.z;
      s.w += t.w;
    }
    if (bias != nullptr) {
      const uint2 b = *reinterpret_cast<const uint2*>(bias + n);
      s.x += bf16lo(b.x);
      s.y += bf16hi(b.x);
      s.z += bf16lo(b.y);
      s.w += bf16hi(b.y);
    }
    uint2 v = make_uint2(pack_bf16x2(s.x, s.y), pack_bf16x2(s.z, s.w));
    if (rowbias != nullptr) {
      const uint2 r = *reinterpret_cast<const uint2*>(rowbias + (m / rows_per_sample) * rowbias_ld + n);
      v.x = pack_bf16x2(bf16lo(v.x) + bf16lo(r.x), bf16hi(v.x) + bf16hi(r.x));
      v.y = pack_bf16x2(bf16lo(v.y) + bf16lo(r.y), bf16hi(v.y) + bf16hi(r.y));
    }
    if (residual != nullptr) {
      const uint2 r = *reinterpret_cast<const uint2*>(residual + m * ldr + n);
      v.x = pack_bf16x2(bf16lo(v.x) + bf16lo(r.x), bf16hi(v.x) + bf16hi(r.x));
      v.y = pack_bf16x2(bf16lo(v.y) + bf16lo(r.y), bf16hi(v.y) + bf16hi(r.y));
    }
    *reinterpret_cast<uint2*>(C + m * ldc + n) = v;
  }
}

// C[m][n] (fp32) += alpha * sum_z slab[z][m][n]   (weight-gradient accumulation after a split-K NT GEMM)
__global__ __launch_bounds__(256) void splitk_accum_kernel(const float* __restrict__ slabs, int splits, long M, int N,
                                                           float alpha, float* __restrict__ C, long ldc) {
  const long nchunk = M * (N / 4);
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < nchunk; id += (long)gridDim.x * blockDim.x) {
    const long m = id / (N / 4);
    const int n = (int)(id - m * (N / 4)) * 4;
    float4 s = *reinterpret_cast<const float4*>(slabs + m * N + n);
    for (int z = 1; z < splits; ++z) {
      const float4 t = *reinterpret_cast<const float4*>(slabs + ((long)z * M + m) * N + n);
      s.x += t.x, s.y += t.y, s.z += t.z, s.w += t.w;
    }
    float* c = C + m * ldc + n;
    c[0] += alpha * s.x, c[1] += alpha * s.y, c[2] += alpha * s.z, c[3] += alpha * s.w;
  }
}

// dst[c][r] = src[r][c]  (bf16, rows x cols with leading dimension ld -> cols x rows, dense); 64x64 tiles through LDS,
// 16-byte global accesses on both sides
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* __restrict__ src, long rows, int cols, long ld,
                                                             bf16_t* __restrict__ dst) {
  __shared__ bf16_t tile[64][64 + 8];
  const long r0 = (long)blockIdx.x * 64;
  const int c0 = blockIdx.y * 64;
  const int tr = threadIdx.x >> 3, tc = (threadIdx.x & 7) * 8;  // 32 rows x 8 chunks per pass
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const long r = r0 + tr + 32 * p;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (r < rows && c0 + tc < cols) v = *reinterpret_cast<const uint4*>(src + r * ld + c0 + tc);
    *reinterpret_cast<uint4*>(&tile[tr + 32 * p][tc]) = v;
  }
  __syncthreads();
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int c = tr + 32 * p;  // output row = source column
    if (c0 + c >= cols) continue;
    bf16_t o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = tile[tc + e][c];
    const long r = r0 + tc;
    if (r + 8 <= rows) {
      *reinterpret_cast<uint4*>(dst + (long)(c0 + c) * rows + r) = *reinterpret_cast<const uint4*>(o);
    } else {
      for (int e = 0; e < 8 && r + e < rows; ++e) dst[(long)(c0 + c) * rows + r + e] = o[e];
    }
  }
}

struct OutSpec {
  const bf16_t* bias;
  const bf16_t* rowbias;
  long rowbias_ld;
  int rows_per_sample;
  const bf16_t* residual;
  long ldr;
  bf16_t* C;
  long ldc;
  bf16_t* C2;
  long ldc2;
  const bf16_t* rowscale;
  bf16_t* G = nullptr;  // GEGLU epilogue (EpiParams::geglu_F): activated output [M][geglu_F]
  long ldg = 0;
  int geglu_F = 0;
  int c_row0 = 0;
  const bf16_t* gb_h = nullptr;  // GEGLU-backward epilogue (EpiParams::gb_F = N): saved pre-activation [M][2N]
  long gb_ldh = 0;
  int res_mod = 0;               // EpiParams::res_mod
  int* defer_splits = nullptr;   // not null: a split-K launch leaves its fp32 slabs UNFINISHED in ws and reports the split count here
                                 // (1 = the output is complete); the caller's next launch consumes the slabs (aql_groupnorm_silu_*_slabs)
};

inline int pick_splits(int tiles, int kt_total, long M, int N, size_t ws_bytes) {
  // split-K pays only when K is deep (the fp32 slabs cost 8 B per output element per split) and the grid is small
  static const int deep_kt = AQL_TUNE_INT("AQL_DEEPKT", 32);  // tuning hook
  if (tiles >= 224 || kt_total < deep_kt) return 1;
  int s = (320 + tiles - 1) / tiles;
  if (s > kt_total / 8) s = kt_total / 8;
  if (s > 16) s = 16;
  while (s > 1 && (size_t)s * (size_t)M * (size_t)N * 4u > ws_bytes) --s;
  return s < 1 ? 1 : s;
}

static int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

// Tile configurations.  ids 0-4: the transposing (token-reduction) kernels on the 32x32x16 MFMA; ids 5-10: the
// pipelined buffer-load kernels on the 16x16x32 MFMA used by every bf16-output GEMM / conv.
enum { P_128x160 = 5, P_64x160 = 6, P_32x160 = 7, P_64x64 = 8, P_128x32 = 9, P_128x128 = 10, P_W128x160 = 11, P_W64x160 = 12, P_W32x160 = 13,
       P_W256x160 = 14, P_W256x160B = 15, P_W128x160L8 = 16, P_256x256 = 17 };   // 17 (AQL_TILE only, -DAQL_T256 build): 256x256 on FOUR wavefronts of 128x128, one per SIMD, no loader wavefronts   // experiments (-DAQL_BIGWAVE, AQL_TILE only): 15 = 256x160 on FOUR compute wavefronts of 128x80, 16 = 128x160 with EIGHT loader wavefronts

template <class LA, class LB, int EPI>
void launch_cfg(int cfg, int pd, const GemmArgs<LA, LB>& g, hipStream_t stream) {
  if constexpr (LA::kTrans) {
    switch (cfg) {
      case 0: launch_gemm<128, 32, 32, 32, LA, LB, EPI>(g, stream); break;
      case 1: launch_gemm<128, 128, 64, 64, LA, LB, EPI>(g, stream); break;
      case 2: launch_gemm<128, 64, 64, 32, LA, LB, EPI>(g, stream); break;
      default: launch_gemm<64, 64, 32, 32, LA, LB, EPI>(g, stream); break;
    }
  } else {
// pd: 12 = LDS-DMA ring with two stages; 13 = as many stages as fit the 160 KB LDS (DEEP), for grids of at most one
// workgroup per CU
#define AQL_P(BM, BN, WM, WN, PDHI, DEEP)                                                        \
  if (pd == 13) return launch_gemm_d<BM, BN, WM, WN, LA, LB, EPI, DEEP>(g, stream);              \
  return launch_gemm_d<BM, BN, WM, WN, LA, LB, EPI, 2>(g, stream);
    if constexpr (std::is_same<LA, ConvFwdLoader>::value && EPI == EPI_BF16) {  // ablation probes, never in production
      static const int abl = AQL_TUNE_INT("AQL_ABL", 0);
      if (abl && cfg == P_64x160) {
        if (abl == 1) return launch_gemm_d<64, 160, 32, 80, LA, LB, EPI, 2, 1>(g, stream);
        if (abl == 2) return launch_gemm_d<64, 160, 32, 80, LA, LB, EPI, 2, 2>(g, stream);
        if (abl == 3) return launch_gemm_d<64, 160, 32, 80, LA, LB, EPI, 2, 3>(g, stream);
        if (abl == 9) return launch_gemm_d<64, 160, 32, 80, LA, LB, EPI, 2, 9>(g, stream);
        return launch_gemm_d<64, 160, 32, 80, LA, LB, EPI, 2, 4>(g, stream);
      }
    }
    if constexpr ((std::is_same<LA, ConvFwdLoader>::value || std::is_same<LA, ConvBwdLoader>::value) &&
                  std::is_same<LB, PlainLoader>::value && (EPI == EPI_BF16 || EPI == EPI_SLAB)) {
      // stride-1 convs on 64- / 32-pixel-wide maps: the row-tile form (one A tile per (kh, channel slab) serves the three kw taps),
      // forward and backward-data, on the 256-row tile (one or two whole rounds) and on the 128-row wave-specialised tile
      static const int conv_row = AQL_TUNE_INT("AQL_CONV_ROW", 1);                 // A/B hook (0 = off, 1 = all, 2 = only the 256-row tile)
      static const int conv_row_rounds = AQL_TUNE_INT("AQL_CONV_ROW_ROUNDS", 2);   // 256-row tile: grids of up to this many whole rounds
      const int t256 = (g.M / 256) * aql_cdiv(g.N, 160);
      const bool r256 = cfg == P_W256x160 || (std::is_same<LA, ConvFwdLoader>::value && g.M % 256 == 0 && t256 % 256 == 0 &&
                                              t256 / 256 <= conv_row_rounds);
      // the VAE's wide maps (128 / 256 pixels per row, 128-multiples of channels): many rounds of 256-pixel row tiles (AQL_CONV_ROW_WIDE=0: off)
      static const int conv_row_wide = AQL_TUNE_INT("AQL_CONV_ROW_WIDE", 1);
      const bool wide = conv_row_wide && g.splits == 1 && g.M % 256 == 0 && (g.a0.Win == 128 || g.a0.Win == 256 || (g.a0.Win == 512 && g.N % 128 == 0 && g.N % 160 != 0)) && g.M / 256 >= 256;
      if (conv_row && (r256 || wide) && aqlconvrow::try_conv_row<LA, EPI>(g, 256, stream)) return;
      if (conv_row == 1 && cfg == P_W128x160 && g.M % 128 == 0 && aqlconvrow::try_conv_row<LA, EPI>(g, 128, stream)) return;
    }
#ifdef AQL_T256
    if constexpr (std::is_same<LA, PlainLoader>::value && std::is_same<LB, PlainLoader>::value && EPI == EPI_BF16) {
      if (cfg == P_256x256) return launch_gemm_d<256, 256, 128, 128, LA, LB, EPI, 2>(g, stream);
    }
#endif
    if (cfg == P_W256x160) return launch_gemm_w<256, 160, 64, 80, LA, LB, EPI, 3, 8>(g, stream);
#ifdef AQL_BIGWAVE
    if (cfg == P_W256x160B) return launch_gemm_w<256, 160, 128, 80, LA, LB, EPI, 3, 4>(g, stream);
    if (cfg == P_W128x160L8) return launch_gemm_w<128, 160, 64, 80, LA, LB, EPI, 4, 4, 8>(g, stream);
#endif
    if (cfg == P_W128x160) return launch_gemm_w<128, 160, 64, 80, LA, LB, EPI, 4>(g, stream);
    if (cfg == P_W64x160) return launch_gemm_w<64, 160, 32, 80, LA, LB, EPI, 5>(g, stream);
    if (cfg == P_W32x160) return launch_gemm_w<32, 160, 16, 80, LA, LB, EPI, 6>(g, stream);
    switch (cfg) {
      case P_128x160: { AQL_P(128, 160, 64, 80, 3, 4) }
      case P_64x160: { AQL_P(64, 160, 32, 80, 4, 5) }
      case P_32x160: { AQL_P(32, 160, 16, 80, 4, 6) }
      case P_128x32: { AQL_P(128, 32, 32, 32, 4, 6) }
      case P_128x128: { AQL_P(128, 128, 64, 64, 3, 4) }
      default: { AQL_P(64, 64, 32, 32, 4, 8) }
    }
#undef AQL_P
  }
}

// tile choice for the token-reduction kernels: the largest tile that still yields >= ~1 workgroup per CU
inline int pick_cfg(long M, int N, int kt_total, bool can_split, int* tiles) {
  if (N <= 32) {
    *tiles = aql_cdiv(M, 128);
    return 0;
  }
  const int t128 = aql_cdiv(M, 128) * aql_cdiv(N, 128);
  const int t64 = aql_cdiv(M, 128) * aql_cdiv(N, 64);
  const int t6464 = aql_cdiv(M, 64) * aql_cdiv(N, 64);
  const bool waste128 = (N % 128 != 0) && (N % 128 <= 64);
  if (t128 >= 240 && !waste128) {
    *tiles = t128;
    return 1;
  }
  if (t64 >= 240 || (waste128 && t128 >= 240)) {
    *tiles = t64;
    return 2;
  }
  *tiles = t6464;
  return 3;
}

// Tile + prefetch depth for a bf16-output GEMM.  Every channel count of the U-Net is a multiple of 160, so the 160-wide
// tiles cover the model; the 64x64 / 128x128 / 128x32 tiles take the odd shapes (LoRA rank, tests).  Measured per shape
// with tools/probe_gemm.py on MI355X.
inline void pick_tile(long M, int N, int kt_total, bool can_split, int* cfg, int* tiles, int* pd) {
  static const int force = env_int("AQL_TILE", 0), force_pd = AQL_TUNE_INT("AQL_PD", 0);  // tuning hooks
  static const int deep_kt = AQL_TUNE_INT("AQL_DEEPKT", 32);  // tuning hook
  const bool deep = can_split && kt_total >= deep_kt;
  if (N % 160 == 0) {
    const int nt = N / 160;
    const int t128 = aql_cdiv(M, 128) * nt, t64 = aql_cdiv(M, 64) * nt, t32 = aql_cdiv(M, 32) * nt;
    // (round 4, tools/cmp_vendor.py GEMM_SHAPES sweep on the guided-sampling shapes: 512 x 3840 x 1280 runs 19.5 us on 192 tiles of
    // 64 x 160 against 29.0 on 384 of 32 x 160; 512 x 1280 x 1280 13.2 us on 160 tiles of 64 x 64 against 16.3 on 128 of 32 x 160)
    if (t128 >= 448 || (deep && t64 < 448)) *cfg = P_128x160, *tiles = t128;
    else if (t64 >= 160) *cfg = P_64x160, *tiles = t64;
    else if (t32 >= 192) *cfg = P_32x160, *tiles = t32;
    else *cfg = P_64x64, *tiles = aql_cdiv(M, 64) * aql_cdiv(N, 64);
    // wave-specialised kernels (one 8-wave workgroup per CU): they win when the grid is a whole number of chip-wide
    // rounds and K is long enough to amortise the un-overlapped prologue / epilogue (measured, tools/tune_gemm.py)
    static const int use_w = AQL_TUNE_INT("AQL_W", 1);
    if (use_w && kt_total >= 8) {
      if (t128 >= 240 && t128 <= 768) *cfg = P_W128x160, *tiles = t128;
      else if (deep && t128 < 240 && use_w != 3) *cfg = P_W128x160, *tiles = t128;  // split K up to one chip-wide round
      else if (t128 < 240 && t64 >= 240 && t64 <= 512) *cfg = P_W64x160, *tiles = t64;
      else if (t64 < 160 && t32 >= 240 && t32 <= 512) *cfg = P_W32x160, *tiles = t32;
    }
    if (force >= P_128x160 && force <= P_32x160) {
      *cfg = force;
      *tiles = aql_cdiv(M, force == P_128x160 ? 128 : force == P_64x160 ? 64 : 32) * nt;
    }
    // 12-wave workgroups on 256x160 tiles: ONE chip-wide round (AQL_W256=0 disables)
    static const int use_w256 = AQL_TUNE_INT("AQL_W256", 1);
    const int t256 = aql_cdiv(M, 256) * nt;
    // (round 5: also 176-239 tiles under a short K -- q | k | v at the 16 x 16 / 32 x 32 levels, 2048 x 3840 x 1280 and 4096 x 1920 x 640:
    // 192 tiles in ONE round on three quarters of the chip, 34.2 / 22.4 us against 41.5 / 24.8 on one and a half rounds of 128 x 160;
    // at K >= 2560 the same grid loses, 88 against 63 us)
    if (use_w && use_w256 && kt_total >= 8 && ((t256 >= 240 && t256 <= 256) || (t256 >= 176 && t256 < 240 && kt_total <= 20)))
      *cfg = P_W256x160, *tiles = t256;
    if (force == P_W256x160 || force == P_W256x160B) *cfg = force, *tiles = t256;
    if (force == P_W128x160 || force == P_W128x160L8) *cfg = force, *tiles = t128;
    if (force == P_W64x160) *cfg = force, *tiles = t64;
    if (force == P_W32x160) *cfg = force, *tiles = t32;
  } else if (N <= 32) {
    *cfg = P_128x32, *tiles = aql_cdiv(M, 128);
  } else {
    const int t128 = aql_cdiv(M, 128) * aql_cdiv(N, 128);
    if (t128 >= 240 || deep) *cfg = P_128x128, *tiles = t128;
    else *cfg = P_64x64, *tiles = aql_cdiv(M, 64) * aql_cdiv(N, 64);
  }
  if (force == P_64x64) *cfg = P_64x64, *tiles = aql_cdiv(M, 64) * aql_cdiv(N, 64);
  if (force == P_256x256) *cfg = P_256x256, *tiles = aql_cdiv(M, 256) * aql_cdiv(N, 256);
  *pd = force_pd ? force_pd : 0;  // 0: chosen from the grid size once the split count is known
}

// Dispatch one bf16-output GEMM over the tile configurations; split-K slabs + finalize when the grid cannot fill
// 256 CUs, K is deep and the caller supplied a workspace.
// bytes an operand spans from its base pointer: the kernels read operands through buffer descriptors of BUF_BYTES (1 GiB) whose
// out-of-range offsets zero-fill -- an operand reaching past that would silently read zeros (the VAE's 256-channel 512 x 512 map at
// batch 16 is 2.1 GiB), so the entry points refuse it and the Python wrappers go through such maps in sample / row chunks
inline long loader_span(const PlainLoader& l) { return (long)l.rows * l.ld * 2; }
inline long loader_span(const ConvFwdLoader& l) { return (long)l.B * l.Hin * l.Win * l.Cin * 2; }
inline long loader_span(const ConvBwdLoader& l) { return (long)l.B * l.Hout * l.Wout * l.Cout * 2; }

template <class LA, class LB>
int run_bf16_gemm(GemmArgs<LA, LB> g, const OutSpec& o, float* ws, size_t ws_bytes, hipStream_t stream,
                  const char* name) {
  {
    const long sa = loader_span(g.a0), sb = loader_span(g.b0);
    const long sa1 = g.ktiles1 > 0 ? loader_span(g.a1) : 0, sb1 = g.ktiles1 > 0 ? loader_span(g.b1) : 0;
    AQL_CHECK_ARG(sa < (long)BUF_BYTES && sb < (long)BUF_BYTES && sa1 < (long)BUF_BYTES && sb1 < (long)BUF_BYTES,
                  "%s: an operand spans %ld bytes, the buffer descriptors cover %u (split the batch / the rows: ops.span_chunks)", name,
                  sa > sb ? sa : sb, BUF_BYTES);
  }
  g.epi = EpiParams{};
  g.epi.rows_per_sample = o.rows_per_sample > 0 ? o.rows_per_sample : 1;
  const int kt_total = g.ktiles0 + g.ktiles1;
  // adjacent tiles share the LARGER operand: the weight panel when it outweighs the activations (deep 3x3 convs at
  // 8x8 / 16x16 / 32x32), else the activation rows
  g.m_fast = ((long)g.N * kt_total > (long)g.M * (kt_total < 9 ? kt_total : kt_total / 9 + 1)) ? 1 : 0;
#ifdef AQL_EXPERIMENTS
  if (const char* e = getenv("AQL_MFAST")) g.m_fast = atoi(e);
#endif
  int tiles = 0, cfg = 0, pd = 1;
  pick_tile(g.M, g.N, kt_total, ws != nullptr && o.C2 == nullptr, &cfg, &tiles, &pd);
  if (o.geglu_F > 0 && !(cfg == P_128x160 || cfg == P_64x160 || cfg == P_32x160 || cfg == P_W128x160 || cfg == P_W64x160 ||
                         cfg == P_W32x160))
    return AQL_NOT_FUSED;  // the [80 value | 80 gate] tile layout exists for the 160-wide tiles only
  int splits = 1;
  if (ws != nullptr && o.C2 == nullptr) splits = pick_splits(tiles, kt_total, g.M, g.N, ws_bytes);
  if ((cfg == P_W128x160 || cfg == P_W128x160L8 || cfg == P_W64x160 || cfg == P_W32x160) && splits > 1) {
    // one workgroup per CU: aim at exactly one (or two) chip-wide rounds
    int s2 = 256 / tiles;
    if (s2 < 1) s2 = 1;
    while (s2 > 1 && (kt_total / s2 < 8 || (size_t)s2 * (size_t)g.M * (size_t)g.N * 4u > ws_bytes)) --s2;
    splits = s2;
  }
#ifdef AQL_EXPERIMENTS
  if (ws != nullptr && o.C2 == nullptr) {   // tuning hook (tools/experiments/time_conv8.py), re-read on every call
    if (const char* e = getenv("AQL_SPLITS")) {
      int s = atoi(e);
      while (s > 1 && (kt_total / s < 2 || (size_t)s * (size_t)g.M * (size_t)g.N * 4u > ws_bytes)) --s;
      if (s >= 1) splits = s;
    }
  }
#endif
  if ((o.gb_h != nullptr || o.res_mod > 0) && splits > 1) return AQL_NOT_FUSED;   // the slab + finalize path has neither epilogue
  g.splits = splits;
  // LDS-DMA staging everywhere (measured fastest on every shape, hot or cold operands); a grid of <= 1 workgroup per CU
  // cannot hide latency with occupancy, so it gets the deep stage ring instead
  if (pd == 0) pd = (tiles * splits <= 288) ? 13 : 12;
  if (splits > 1) {
    g.epi.Cf = ws;
    g.epi.ldcf = g.N;
    launch_cfg<LA, LB, EPI_SLAB>(cfg, pd, g, stream);
    AQL_CHECK_LAUNCH(name);
    if (o.defer_splits != nullptr) {   // the finalize is the consumer's first pass
      *o.defer_splits = splits;
      return AQL_OK;
    }
    const long nchunk = (long)g.M * (g.N / 4);
    int blocks = (int)((nchunk + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(splitk_finalize_kernel, dim3(blocks), dim3(256), 0, stream, ws, splits, (long)g.M, g.N,
                       o.bias, o.rowbias, o.rowbias_ld > 0 ? o.rowbias_ld : (long)g.N, g.epi.rows_per_sample, o.residual, o.ldr, o.C,
                       o.ldc);
    AQL_CHECK_LAUNCH(name);
    return AQL_OK;
  }
  g.epi.Cf = ws;  // only read by the ablation/trace probes
  g.epi.C = o.C;
  g.epi.ldc = o.ldc;
  g.epi.bias = o.bias;
  g.epi.residual = o.residual;
  g.epi.ldr = o.ldr;
  g.epi.C2 = o.C2;
  g.epi.ldc2 = o.ldc2;
  g.epi.rowscale = o.rowscale;
  g.epi.rowbias = o.rowbias;
  g.epi.rowbias_ld = o.rowbias_ld > 0 ? o.rowbias_ld : (long)g.N;
  g.epi.G = o.G;
  g.epi.ldg = o.ldg;
  g.epi.geglu_F = o.geglu_F;
  g.epi.c_row0 = o.c_row0;
  g.epi.res_mod = o.res_mod;
  if (o.gb_h != nullptr) g.epi.gb_h = o.gb_h, g.epi.gb_ldh = o.gb_ldh, g.epi.gb_F = g.N;
  launch_cfg<LA, LB, EPI_BF16>(cfg, pd, g, stream);
  AQL_CHECK_LAUNCH(name);
  return AQL_OK;
}

inline PlainLoader plain(const bf16_t* p, long ld, long rows, int K) {
  PlainLoader l;
  l.base = p;
  l.ld = ld;
  l.rows = (int)rows;
  l.K = K;
  return l;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
extern "C" int aql_gemm_bf16_ex(const bf16_t* A, long lda, const bf16_t* B, long ldb, long M, int N, int K,
                                const bf16_t* A2, long lda2, const bf16_t* B2, long ldb2, int K2, const bf16_t* bias,
                                const bf16_t* rowbias, int rows_per_sample, const bf16_t* residual, long ldr,
                                bf16_t* C, long ldc, long lora_row0, float* ws, size_t ws_bytes, hipStream_t stream);

extern "C" int aql_gemm_bf16(const bf16_t* A, long lda, const bf16_t* B, long ldb, long M, int N, int K,
                             const bf16_t* A2, long lda2, const bf16_t* B2, long ldb2, int K2, const bf16_t* bias,
                             const bf16_t* rowbias, int rows_per_sample, const bf16_t* residual, long ldr,
                             bf16_t* C, long ldc, float* ws, size_t ws_bytes, hipStream_t stream) {
  return aql_gemm_bf16_ex(A, lda, B, ldb, M, N, K, A2, lda2, B2, ldb2, K2, bias, rowbias, rows_per_sample, residual, ldr, C, ldc,
                          0, ws, ws_bytes, stream);
}

// aql_gemm_bf16 for twin batches: rows below lora_row0 (the clean half, all-zero scale rows) have no second-K-segment term --
// output tiles that end at or below it skip the segment, a straddling tile reads those rows of A2 as zeros.
extern "C" int aql_gemm_bf16_ex(const bf16_t* A, long lda, const bf16_t* B, long ldb, long M, int N, int K,
                                const bf16_t* A2, long lda2, const bf16_t* B2, long ldb2, int K2, const bf16_t* bias,
                                const bf16_t* rowbias, int rows_per_sample, const bf16_t* residual, long ldr,
                                bf16_t* C, long ldc, long lora_row0, float* ws, size_t ws_bytes, hipStream_t stream) {
  AQL_CHECK_ARG(A && B && C, "aql_gemm_bf16: null operand");
  AQL_CHECK_ARG(M > 0 && N > 0 && K > 0 && M < (1L << 31), "aql_gemm_bf16: bad shape M=%ld N=%d K=%d", M, N, K);
  AQL_CHECK_ARG(N % 8 == 0 && K % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0,
                "aql_gemm_bf16: N,K and leading dims must be multiples of 8 (N=%d K=%d)", N, K);
  AQL_CHECK_ARG(aligned16(A) && aligned16(B) && aligned16(C), "aql_gemm_bf16: pointers must be 16-byte aligned");
  AQL_CHECK_ARG(residual == nullptr || (ldr % 8 == 0 && aligned16(residual)), "aql_gemm_bf16: bad residual");
  GemmArgs<PlainLoader, PlainLoader> g;
  g.a0 = plain(A, lda, M, K);
  g.b0 = plain(B, ldb, N, K);
  g.ktiles0 = aql_cdiv(K, BK);
  g.ktiles1 = 0;
  g.a1 = g.a0;
  g.b1 = g.b0;
  if (A2 != nullptr) {
    AQL_CHECK_ARG(B2 && K2 > 0 && K2 % 8 == 0 && lda2 % 8 == 0 && ldb2 % 8 == 0 && aligned16(A2) && aligned16(B2),
                  "aql_gemm_bf16: bad second K segment");
    g.a1 = plain(A2, lda2, M, K2);
    g.b1 = plain(B2, ldb2, N, K2);
    g.ktiles1 = aql_cdiv(K2, BK);
    if (lora_row0 > 0) {   // twin batch: rows below lora_row0 have no second-segment (LoRA) term
      g.seg1_row0 = (int)(lora_row0 > M ? M : lora_row0);
      g.a1.row_lo = g.seg1_row0;
    }
  }
  g.M = (int)M;
  g.N = N;
  OutSpec o{bias, rowbias, 0, rows_per_sample, residual, ldr, C, ldc, nullptr, 0, nullptr};
  return run_bf16_gemm(g, o, ws, ws_bytes, stream, "aql_gemm_bf16");
}

// Column-grouped form of aql_gemm_bf16_ex (round 6): output columns [g grp_n, (g + 1) grp_n) read A at a column offset of g grp_a
// elements and A2 at g grp_a2 (0 = the same columns for every group); B / B2 rows are the output columns as always.  Block-diagonal
// products as ONE launch -- the rank-320 LoRA of q | k | v (BASELINE config 3; utils/lora_modules.py:9-26 on three linears that
// read the same tokens, scripts/lib/original_unet.py:688-704):
//   forward    [q | k | v] = X.[Wq; Wk; Wv]^T + Ts_g.Bup_g^T        A2 = [Ts_q | Ts_k | Ts_v] (grp_a2 = r), B2 = [Bup_q; Bup_k; Bup_v]
//   backward   [dTs_q | dTs_k | dTs_v] = dY_g.Bup_g                 A = [dQ | dK | dV] (grp_a = C), B = [Bup_q^T; Bup_k^T; Bup_v^T]
// Optional second output C2 = C * rowscale[m / rows_per_sample][n] (the per-message scale: Ts = T * S; then no split K).
// grp_n: a multiple of 320 (every tile width the picker chooses divides it) that divides N; N a multiple of 160.
extern "C" int aql_gemm_bf16_grouped(const bf16_t* A, long lda, const bf16_t* B, long ldb, long M, int N, int K, const bf16_t* A2,
                                     long lda2, const bf16_t* B2, long ldb2, int K2, int grp_n, int grp_a, int grp_a2,
                                     const bf16_t* bias, const bf16_t* residual, long ldr, bf16_t* C, long ldc, bf16_t* C2, long ldc2,
                                     const bf16_t* rowscale, int rows_per_sample, long lora_row0, float* ws, size_t ws_bytes,
                                     hipStream_t stream) {
  AQL_CHECK_ARG(A && B && C, "aql_gemm_bf16_grouped: null operand");
  AQL_CHECK_ARG(M > 0 && N > 0 && K > 0 && M < (1L << 31), "aql_gemm_bf16_grouped: bad shape M=%ld N=%d K=%d", M, N, K);
  AQL_CHECK_ARG(N % 160 == 0 && K % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0,
                "aql_gemm_bf16_grouped: N must be a multiple of 160, K and the leading dimensions of 8 (N=%d K=%d)", N, K);
  AQL_CHECK_ARG(grp_n > 0 && grp_n % 320 == 0 && N % grp_n == 0 && grp_a >= 0 && grp_a2 >= 0 && grp_a % 8 == 0 && grp_a2 % 8 == 0,
                "aql_gemm_bf16_grouped: grp_n = %d must be a multiple of 320 that divides N = %d; offsets multiples of 8", grp_n, N);
  AQL_CHECK_ARG(aligned16(A) && aligned16(B) && aligned16(C), "aql_gemm_bf16_grouped: pointers must be 16-byte aligned");
  AQL_CHECK_ARG(residual == nullptr || (ldr % 8 == 0 && aligned16(residual)), "aql_gemm_bf16_grouped: bad residual");
  AQL_CHECK_ARG(C2 == nullptr || (rowscale != nullptr && ldc2 % 8 == 0 && aligned16(C2) && rows_per_sample > 0),
                "aql_gemm_bf16_grouped: the second output needs rowscale [M / rows_per_sample][N]");
  const int ng = N / grp_n;
  AQL_CHECK_ARG(lda >= (long)(ng - 1) * grp_a + K, "aql_gemm_bf16_grouped: A rows hold %ld elements, the groups read %ld", lda,
                (long)(ng - 1) * grp_a + K);
  GemmArgs<PlainLoader, PlainLoader> g;
  g.a0 = plain(A, lda, M, K);
  g.b0 = plain(B, ldb, N, K);
  g.ktiles0 = aql_cdiv(K, BK);
  g.ktiles1 = 0;
  g.a1 = g.a0;
  g.b1 = g.b0;
  if (A2 != nullptr) {
    AQL_CHECK_ARG(B2 && K2 > 0 && K2 % 8 == 0 && lda2 % 8 == 0 && ldb2 % 8 == 0 && aligned16(A2) && aligned16(B2) &&
                      lda2 >= (long)(ng - 1) * grp_a2 + K2,
                  "aql_gemm_bf16_grouped: bad second K segment");
    g.a1 = plain(A2, lda2, M, K2);
    g.b1 = plain(B2, ldb2, N, K2);
    g.ktiles1 = aql_cdiv(K2, BK);
    if (lora_row0 > 0) {
      g.seg1_row0 = (int)(lora_row0 > M ? M : lora_row0);
      g.a1.row_lo = g.seg1_row0;
    }
  }
  g.M = (int)M;
  g.N = N;
  g.grp_n = grp_n;
  g.grp_a0 = grp_a;
  g.grp_a1 = grp_a2;
  OutSpec o{bias, nullptr, 0, rows_per_sample > 0 ? rows_per_sample : 1, residual, ldr, C, ldc, C2, ldc2, rowscale};
  return run_bf16_gemm(g, o, C2 == nullptr ? ws : nullptr, C2 == nullptr ? ws_bytes : 0, stream, "aql_gemm_bf16_grouped");
}

// Backward-data of ff.net.2 in the two-launch LoRA form (any rank) fused with the backward of the GEGLU in front of it
// (original_unet.py:727-729): d(activated) [M][F] = A.B^T + A2.B2^T is turned into d(pre-activation) DH [M][2F] by the epilogue
// (aql_geglu_bwd on the bf16-rounded tile, saved pre-activation H [M][2F]).  Returns AQL_NOT_FUSED (100) when the shape would
// take the split-K path: the caller then runs the plain GEMM and aql_geglu_bwd.
extern "C" int aql_gemm_bf16_geglu_bwd(const bf16_t* A, long lda, const bf16_t* B, long ldb, long M, int F, int K,
                                       const bf16_t* A2, long lda2, const bf16_t* B2, long ldb2, int K2, const bf16_t* H,
                                       long ldh, bf16_t* DH, long lddh, float* ws, size_t ws_bytes, hipStream_t stream) {
  AQL_CHECK_ARG(A && B && H && DH, "aql_gemm_bf16_geglu_bwd: null operand");
  AQL_CHECK_ARG(M > 0 && F > 0 && K > 0 && M < (1L << 31) && F % 8 == 0 && K % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 &&
                    ldh % 8 == 0 && lddh % 8 == 0 && ldh >= 2 * F && lddh >= 2 * F,
                "aql_gemm_bf16_geglu_bwd: bad shape M=%ld F=%d K=%d", M, F, K);
  AQL_CHECK_ARG(aligned16(A) && aligned16(B) && aligned16(H) && aligned16(DH), "aql_gemm_bf16_geglu_bwd: pointers must be 16-byte aligned");
  GemmArgs<PlainLoader, PlainLoader> g;
  g.a0 = plain(A, lda, M, K);
  g.b0 = plain(B, ldb, F, K);
  g.ktiles0 = aql_cdiv(K, BK);
  g.ktiles1 = 0;
  g.a1 = g.a0;
  g.b1 = g.b0;
  if (A2 != nullptr) {
    AQL_CHECK_ARG(B2 && K2 > 0 && K2 % 8 == 0 && lda2 % 8 == 0 && ldb2 % 8 == 0 && aligned16(A2) && aligned16(B2),
                  "aql_gemm_bf16_geglu_bwd: bad second K segment");
    g.a1 = plain(A2, lda2, M, K2);
    g.b1 = plain(B2, ldb2, F, K2);
    g.ktiles1 = aql_cdiv(K2, BK);
  }
  g.M = (int)M;
  g.N = F;
  OutSpec o{nullptr, nullptr, 0, 1, nullptr, 0, DH, lddh, nullptr, 0, nullptr};
  o.gb_h = H;
  o.gb_ldh = ldh;
  return run_bf16_gemm(g, o, ws, ws_bytes, stream, "aql_gemm_bf16_geglu_bwd");
}

// ff.net.0.proj + GEGLU in one launch (scripts/lib/original_unet.py:727-729 on top of lora_modules.py:56-62):
//   h = A.B^T (+ A2.B2^T) + bias            [M][2F]   (B: [2F][K], rows 0..F-1 "value", F..2F-1 "gate")
//   G = h[:, :F] * gelu_erf(h[:, F:])       [M][F]
// Every 160-wide output tile holds 80 value columns and the matching 80 gate columns (the weight-side loader reads the
// two row ranges), so the activation is applied to the bf16-rounded tile in the epilogue -- bit-identical to the GEMM
// followed by aql_geglu_fwd.  H (the pre-activation, needed by aql_geglu_bwd) is written only when H != null.
// Returns AQL_NOT_FUSED (100) when no 160-wide tile serves the shape (F % 80 != 0, tiny grids): the caller then runs the
// two kernels.
extern "C" int aql_gemm_bf16_geglu(const bf16_t* A, long lda, const bf16_t* B, long ldb, long M, int F, int K,
                                   const bf16_t* A2, long lda2, const bf16_t* B2, long ldb2, int K2, const bf16_t* bias,
                                   bf16_t* H, long ldh, bf16_t* G, long ldg, long lora_row0, hipStream_t stream) {
  AQL_CHECK_ARG(A && B && G, "aql_gemm_bf16_geglu: null operand");
  AQL_CHECK_ARG(M > 0 && F > 0 && K > 0 && M < (1L << 31) && F % 8 == 0 && K % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 &&
                    ldg % 8 == 0 && (H == nullptr || ldh % 8 == 0),
                "aql_gemm_bf16_geglu: bad shape M=%ld F=%d K=%d", M, F, K);
  AQL_CHECK_ARG(aligned16(A) && aligned16(B) && aligned16(G) && aligned16(H), "aql_gemm_bf16_geglu: pointers must be 16-byte aligned");
  if (A2 != nullptr)
    AQL_CHECK_ARG(B2 && K2 > 0 && K2 % 8 == 0 && lda2 % 8 == 0 && ldb2 % 8 == 0 && aligned16(A2) && aligned16(B2),
                  "aql_gemm_bf16_geglu: bad second K segment");
  {   // large grids: the 256 x 256 persistent tile (aql_gemm_lora_t256.cuh), bit-identical to the kernels below
    const int rc = aqlt256::t256_geglu_two_segments(A, lda, B, ldb, M, F, K, A2, lda2, B2, ldb2, K2, bias, H, ldh, G, ldg, lora_row0, stream);
    if (rc != AQL_NOT_FUSED) return rc;
  }
  if (F % 80 != 0) return AQL_NOT_FUSED;
  GemmArgs<PlainLoader, PlainLoader> g;
  g.a0 = plain(A, lda, M, K);
  g.b0 = plain(B, ldb, 2L * F, K);
  g.b0.gsplit = 80;
  g.b0.goff = F - 80;
  g.ktiles0 = aql_cdiv(K, BK);
  g.ktiles1 = 0;
  g.a1 = g.a0;
  g.b1 = g.b0;
  if (A2 != nullptr) {
    AQL_CHECK_ARG(B2 && K2 > 0 && K2 % 8 == 0 && lda2 % 8 == 0 && ldb2 % 8 == 0 && aligned16(A2) && aligned16(B2),
                  "aql_gemm_bf16_geglu: bad second K segment");
    g.a1 = plain(A2, lda2, M, K2);
    g.b1 = plain(B2, ldb2, 2L * F, K2);
    g.b1.gsplit = 80;
    g.b1.goff = F - 80;
    g.ktiles1 = aql_cdiv(K2, BK);
    if (lora_row0 > 0) {
      g.seg1_row0 = (int)(lora_row0 > M ? M : lora_row0);
      g.a1.row_lo = g.seg1_row0;
    }
  }
  g.M = (int)M;
  g.N = 2 * F;
  OutSpec o{bias, nullptr, 0, 1, nullptr, 0, H, ldh, nullptr, 0, nullptr, G, ldg, F};
  o.c_row0 = (int)(lora_row0 < 0 ? 0 : (lora_row0 > M ? M : lora_row0));   // H is only needed where backward runs
  return run_bf16_gemm(g, o, nullptr, 0, stream, "aql_gemm_bf16_geglu");
}

// The general bf16 GEMM with PER-SAMPLE weights -- the weight-side form of the watermark-LoRA linear (utils/lora_modules.py:9-26,
// 56-62): where the rank is not small against the channel count (r = 320 on the 320-channel level: BASELINE config 3) the
// activation-side branch ((x.A^T) * S_b).Bup^T costs 6 M r (N + K) FLOPs per site and sample over forward, backward-data and the
// weight gradients, the effective weight  We_b = W + Bup.diag(S_b).A  built once per sample costs 2 N K r and turns forward and
// backward-data into plain GEMMs:
//     C[m][:] = A[m][:] . Wsel(m)^T (+ bias) (+ residual)     Wsel(m) = B                               for m <  srow0
//                                                                     = Bs + ((m - srow0) / srows) * sstride  otherwise
// (twin batches: rows below srow0 are the clean pass and use the frozen W).  srows (rows per sample) must be a multiple of 256 so
// that no tile straddles two samples; srow0 a multiple of srows.  The same entry BUILDS the per-sample weights: with A = the stacked
// scaled up-matrices [(b, n)][r], B = A_down^T [K][r], residual = W [N][K] and res_mod = N it writes We [(b, n)][K] = W + (Bup*S_b).A.
// geglu_F > 0: ff.net.0 -- B / Bs hold 2 geglu_F rows ([value | gate]), G [M][geglu_F] = value * gelu(gate), C (= H, may be null)
// is written for rows >= c_row0.  gb_h != null: the GEGLU-backward epilogue of aql_gemm_bf16_geglu_bwd (C = DH [M][2N]).
// Returns AQL_NOT_FUSED (100) where the GEGLU forms have no tile.
extern "C" int aql_gemm_bf16_sw(const bf16_t* A, long lda, const bf16_t* B, long ldb, long M, int N, int K, const bf16_t* Bs,
                                long sstride, long srows, long srow0, const bf16_t* bias, const bf16_t* residual, long ldr,
                                long res_mod, bf16_t* C, long ldc, bf16_t* G, long ldg, int geglu_F, long c_row0,
                                const bf16_t* gb_h, long gb_ldh, float* ws, size_t ws_bytes, hipStream_t stream) {
  AQL_CHECK_ARG(A && B && (C || (geglu_F > 0 && G)), "aql_gemm_bf16_sw: null operand");
  AQL_CHECK_ARG(M > 0 && N > 0 && K > 0 && M < (1L << 31) && N % 8 == 0 && K % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 &&
                    (C == nullptr || ldc % 8 == 0), "aql_gemm_bf16_sw: bad shape M=%ld N=%d K=%d", M, N, K);
  AQL_CHECK_ARG(aligned16(A) && aligned16(B) && aligned16(C) && aligned16(Bs) && aligned16(G), "aql_gemm_bf16_sw: pointers must be 16-byte aligned");
  AQL_CHECK_ARG(residual == nullptr || (ldr % 8 == 0 && aligned16(residual) && res_mod >= 0), "aql_gemm_bf16_sw: bad residual");
  AQL_CHECK_ARG(Bs == nullptr || (srows > 0 && srows % 256 == 0 && srow0 >= 0 && srow0 % srows == 0 && sstride % 8 == 0),
                "aql_gemm_bf16_sw: rows per sample (%ld) must be a multiple of 256, srow0 (%ld) a multiple of it", srows, srow0);
  AQL_CHECK_ARG(geglu_F == 0 || (geglu_F > 0 && N == 2 * geglu_F && G != nullptr && ldg % 8 == 0 && residual == nullptr && gb_h == nullptr),
                "aql_gemm_bf16_sw: GEGLU form needs N = 2 F, G and no residual");
  AQL_CHECK_ARG(gb_h == nullptr || (gb_ldh % 8 == 0 && gb_ldh >= 2 * N && aligned16(gb_h) && bias == nullptr), "aql_gemm_bf16_sw: bad GEGLU-backward operand");
  if (geglu_F > 0 && geglu_F % 80 != 0) return AQL_NOT_FUSED;
  GemmArgs<PlainLoader, PlainLoader> g;
  g.a0 = plain(A, lda, M, K);
  g.b0 = plain(B, ldb, N, K);
  if (geglu_F > 0) g.b0.gsplit = 80, g.b0.goff = geglu_F - 80;
  if (Bs != nullptr) {
    g.b0.sbase = Bs;
    g.b0.sstride = sstride;
    g.b0.srows = (int)srows;
    g.b0.s0 = (int)(srow0 / srows);
  }
  g.ktiles0 = aql_cdiv(K, BK);
  g.ktiles1 = 0;
  g.a1 = g.a0;
  g.b1 = g.b0;
  g.M = (int)M;
  g.N = N;
  OutSpec o{bias, nullptr, 0, 1, residual, ldr, C, ldc, nullptr, 0, nullptr};
  o.res_mod = (int)res_mod;
  if (geglu_F > 0) {
    o.G = G, o.ldg = ldg, o.geglu_F = geglu_F;
    o.c_row0 = (int)(c_row0 < 0 ? 0 : (c_row0 > M ? M : c_row0));
  }
  if (gb_h != nullptr) o.gb_h = gb_h, o.gb_ldh = gb_ldh;
  // split K only for the plain form (the slab path knows neither the GEGLU epilogues nor the residual modulo)
  const bool plain_form = geglu_F == 0 && gb_h == nullptr && res_mod == 0;
  return run_bf16_gemm(g, o, plain_form ? ws : nullptr, plain_form ? ws_bytes : 0, stream, "aql_gemm_bf16_sw");
}

// Skinny rank-r "down" GEMM for r <= 64: HBM-bound (reads X once), so no LDS staging of operands.  A workgroup owns
// 16 rows; its 4 wavefronts split K in interleaved 32-wide steps (together they read 256 contiguous bytes per row),
// stream X straight into MFMA B-operands, and the four partial r x 16 accumulators are summed through LDS.
template <int RF>
__device__ __forceinline__ void lora_down_skinny_body(const bf16_t* __restrict__ X, long ldx, long M, int K,
                                                      const bf16_t* __restrict__ A, const bf16_t* __restrict__ S, int rps,
                                                      bf16_t* __restrict__ T, bf16_t* __restrict__ Ts,
                                                      const bf16_t* __restrict__ Tref, float* __restrict__ dS) {
  __shared__ f32x4_t part[4][RF][64];
  // `wave` must be PROVABLY wave-uniform (an SGPR): the `s < nsteps` guards below then compile to scalar branches.  As a plain
  // threadIdx expression hipcc predicated the d = 1 step with EXEC instead (s_and_saveexec, no s_cbranch_execz) -- and MFMA
  // ignores EXEC: a wavefront whose step 4 + wave lies past K accumulated uninitialised registers (NaN for every K < 256;
  // the U-Net's K >= 320 never skip that step, found with a 160-channel test model in round 3)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long row = (long)blockIdx.x * 16 + (lane & 15);
  const int g = lane >> 4;
  const bool ok = row < M;
  // raw buffer loads: rows past M and K steps past the end are out-of-range offsets (hardware zero fill), so there is
  // no branch around any load and all D*(1+RF) loads of an iteration are in flight together
  const __amdgpu_buffer_rsrc_t rx = aqlgemm::make_rsrc(X), ra = aqlgemm::make_rsrc(A);
  const uint32_t xoff = ok ? (uint32_t)(row * ldx + g * 8) * 2u : aqlgemm::OOB_ROW;
  const uint32_t aoff = (uint32_t)((lane & 15) * K + g * 8) * 2u;
  f32x4_t acc[RF];
#pragma unroll
  for (int f = 0; f < RF; ++f) acc[f] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  constexpr int D = 8;  // steps in flight per wavefront
  const int nsteps = K / 32;
  for (int s0 = wave; s0 < nsteps; s0 += 4 * D) {
    aqlgemm::u32x4_t xv[D], av[D][RF];
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int s = s0 + 4 * d;
      if (s < nsteps) {  // wave-uniform: a skipped step issues nothing (an out-of-range load still costs an issue slot)
        const uint32_t so = (uint32_t)s * 64u;
        xv[d] = __builtin_amdgcn_raw_buffer_load_b128(rx, xoff + so, 0, 0);
#pragma unroll
        for (int f = 0; f < RF; ++f)
          av[d][f] = __builtin_amdgcn_raw_buffer_load_b128(ra, aoff + (uint32_t)(f * 16 * K) * 2u + so, 0, 0);
      }
    }
#pragma unroll
    for (int d = 0; d < D; ++d) {
      if (s0 + 4 * d < nsteps) {
#pragma unroll
        for (int f = 0; f < RF; ++f)
          acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8_t*>(&av[d][f]),
                                                           *reinterpret_cast<const bf16x8_t*>(&xv[d]), acc[f], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int f = 0; f < RF; ++f) part[wave][f][lane] = acc[f];
  __syncthreads();
  // wave w finalises fragment(s) f = w, w+4, ...: rows j = lane&15, r index f*16 + g*4 + e
  const int r = RF * 16;
  for (int f = wave; f < RF; f += 4) {
    f32x4_t v = part[0][f][lane];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const f32x4_t u = part[w][f][lane];
      v[0] += u[0]; v[1] += u[1]; v[2] += u[2]; v[3] += u[3];
    }
    const int c = f * 16 + g * 4;
    const uint2 t = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
    if (Tref != nullptr) {
      // backward use: this GEMM produced dTs; dS[sample, c+e] += sum_rows dTs * T  (gradient of the diagonal)
      float pr[4] = {0.f, 0.f, 0.f, 0.f};
      if (ok) {
        const uint2 tr = *reinterpret_cast<const uint2*>(Tref + row * r + c);
        pr[0] = bf16lo(t.x) * bf16lo(tr.x);
        pr[1] = bf16hi(t.x) * bf16hi(tr.x);
        pr[2] = bf16lo(t.y) * bf16lo(tr.y);
        pr[3] = bf16hi(t.y) * bf16hi(tr.y);
      }
      if (rps % 16 == 0) {  // the 16 rows of this workgroup belong to one sample: reduce across them first
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
          for (int o = 1; o < 16; o <<= 1) pr[e] += __shfl_xor(pr[e], o, 64);
        }
        if ((lane & 15) == 0 && (long)blockIdx.x * 16 < M) {
          float* dp = dS + ((long)blockIdx.x * 16 / rps) * r + c;
#pragma unroll
          for (int e = 0; e < 4; ++e) atomicAdd(dp + e, pr[e]);
        }
      } else if (ok) {
        float* dp = dS + (row / rps) * r + c;
#pragma unroll
        for (int e = 0; e < 4; ++e) atomicAdd(dp + e, pr[e]);
      }
    }
    if (!ok) continue;
    *reinterpret_cast<uint2*>(T + row * r + c) = t;
    const uint2 sv = *reinterpret_cast<const uint2*>(S + (row / rps) * r + c);
    *reinterpret_cast<uint2*>(Ts + row * r + c) =
        make_uint2(pack_bf16x2(bf16lo(t.x) * bf16lo(sv.x), bf16hi(t.x) * bf16hi(sv.x)),
                   pack_bf16x2(bf16lo(t.y) * bf16lo(sv.y), bf16hi(t.y) * bf16hi(sv.y)));
  }
}

template <int RF>
__global__ __launch_bounds__(256) void lora_down_skinny_kernel(const bf16_t* __restrict__ X, long ldx, long M, int K,
                                                               const bf16_t* __restrict__ A,
                                                               const bf16_t* __restrict__ S, int rps,
                                                               bf16_t* __restrict__ T, bf16_t* __restrict__ Ts,
                                                               const bf16_t* __restrict__ Tref,
                                                               float* __restrict__ dS) {
  lora_down_skinny_body<RF>(X, ldx, M, K, A, S, rps, T, Ts, Tref, dS);
}

// Up to 32 rank-32 "down" products of equal row count in ONE launch (blockIdx.y = problem): the backward of the text-state
// k|v projections of all cross-attentions, dTs_g = dY_g.Bup_g, dT_g = dTs_g * S  (32 launches of 5-7 us otherwise).
struct DownGroup {
  const bf16_t* X[32];
  const bf16_t* A[32];
  int K[32];
};
__global__ __launch_bounds__(256) void lora_down_skinny_grouped_kernel(const DownGroup g, long M, const bf16_t* __restrict__ S,
                                                                       int rps, bf16_t* __restrict__ T,
                                                                       bf16_t* __restrict__ Ts) {
  const int i = blockIdx.y;
  lora_down_skinny_body<2>(g.X[i], g.K[i], M, g.K[i], g.A[i], S, rps, T + (long)i * M * 32, Ts + (long)i * M * 32, nullptr,
                           nullptr);
}

// T = X.Adown^T (bf16) and Ts = T * S[sample]  -- the rank-r "down" half of the watermark LoRA.
extern "C" int aql_lora_ds(const bf16_t* dTs, const bf16_t* T, int nb, int rows_per_sample, int r, float* dS,
                           hipStream_t stream);

extern "C" int aql_lora_down(const bf16_t* X, long ldx, long M, int K, const bf16_t* Adown, int r, const bf16_t* S,
                             int rows_per_sample, bf16_t* T, bf16_t* Ts, const bf16_t* Tref, float* dS,
                             hipStream_t stream) {
  AQL_CHECK_ARG(X && Adown && S && T && Ts && (Tref == nullptr || dS != nullptr), "aql_lora_down: null operand");
  AQL_CHECK_ARG(M > 0 && r > 0 && r % 8 == 0 && K % 8 == 0 && ldx % 8 == 0 && rows_per_sample > 0,
                "aql_lora_down: bad shape M=%ld r=%d K=%d", M, r, K);
  if (r % 16 == 0 && r <= 64 && K % 32 == 0) {
    const unsigned blocks = (unsigned)((M + 15) / 16);
    switch (r / 16) {
      case 1: hipLaunchKernelGGL(lora_down_skinny_kernel<1>, dim3(blocks), dim3(256), 0, stream, X, ldx, M, K, Adown, S, rows_per_sample, T, Ts, nullptr, nullptr); break;
      case 2: hipLaunchKernelGGL(lora_down_skinny_kernel<2>, dim3(blocks), dim3(256), 0, stream, X, ldx, M, K, Adown, S, rows_per_sample, T, Ts, nullptr, nullptr); break;
      case 3: hipLaunchKernelGGL(lora_down_skinny_kernel<3>, dim3(blocks), dim3(256), 0, stream, X, ldx, M, K, Adown, S, rows_per_sample, T, Ts, nullptr, nullptr); break;
      default: hipLaunchKernelGGL(lora_down_skinny_kernel<4>, dim3(blocks), dim3(256), 0, stream, X, ldx, M, K, Adown, S, rows_per_sample, T, Ts, nullptr, nullptr); break;
    }
    AQL_CHECK_LAUNCH("aql_lora_down");
    if (Tref == nullptr) return AQL_OK;
    return aql_lora_ds(T, Tref, (int)(M / rows_per_sample), rows_per_sample, r, dS, stream);
  }
  GemmArgs<PlainLoader, PlainLoader> g;
  g.a0 = plain(X, ldx, M, K);
  g.b0 = plain(Adown, K, r, K);
  g.a1 = g.a0;
  g.b1 = g.b0;
  g.ktiles0 = aql_cdiv(K, BK);
  g.ktiles1 = 0;
  g.M = (int)M;
  g.N = r;
  OutSpec o{nullptr, nullptr, 0, rows_per_sample, nullptr, 0, T, r, Ts, r, S};
  const int rc = run_bf16_gemm(g, o, nullptr, 0, stream, "aql_lora_down");
  if (rc != AQL_OK || Tref == nullptr) return rc;
  return aql_lora_ds(T, Tref, (int)(M / rows_per_sample), rows_per_sample, r, dS, stream);
}

// Split-K form of the rank-32 down product for few rows under a deep contraction (M <= ~2048, K >= 2048: the backward-data pass
// of ff.net.0 at the 16x16 and 8x8 levels contracts over 10240 features with 1024 / 256 rows; the 16-row workgroups of
// lora_down_skinny_kernel are then too few to fill the chip -- 64 / 16 workgroups: 26 us for 21 MB): the K range is cut into `ks`
// pieces (grid.y), workgroup (row block, piece) reduces its K range and writes its fp32 partial [16][32] to `part` [ks][M][32]; the pieces are then added IN PIECE ORDER (deterministic) and
// T / Ts written -- by a second launch (default: `lora_down_splitk_finalize_kernel`, ordered by the kernel boundary, no
// inter-workgroup communication at all), or, TICKET = true (AQL_DOWN_TICKET=1, a tuning build of the same entry point), by the
// last workgroup of the row block to take a ticket on `counters` (zero before the launch, left zero by it).
//
// The ticket form and the memory model.  Every exchanged word is written with a device-scope relaxed atomic store
// (`global_store ... sc1`: written through to memory, not left dirty in the writing XCD's L2), read with a device-scope relaxed atomic
// load (`global_load ... sc1`: not served from the reading XCD's L2), and the ticket is taken after `s_waitcnt vmcnt(0)`, i.e.
// after the write-through stores have been acknowledged.  That is the code sequence LLVM's AMDGPU memory model itself emits for an
// agent-scope release on gfx942 / gfx950 MINUS the `buffer_wbl2 sc1` that writes back every OTHER dirty line of the L2 (which is
// what makes the fenced form cost 50-65 us here: 512 workgroups each flushing an L2 full of the previous kernels' outputs).  In
// the HIP / C++ model the accesses are all atomic (no data race) but relaxed, so visibility of the pieces to the ticket holder
// is a property of this hardware's sc1 path, not of the language: hence opt-in, guarded to the gfx94x / gfx950 ISAs below, and
// stress-tested (tests/test_gpu_kernels.py: 2000 launches on rotating inputs over stale scratch, bit-compared with the
// two-launch form).
__device__ __forceinline__ void splitk_sum_rows(const float* __restrict__ part, int ks, long M, long rowblock, const bf16_t* __restrict__ S,
                                                int rps, bf16_t* __restrict__ T, bf16_t* __restrict__ Ts, bool coherent) {
  if (threadIdx.x < 128) {   // 16 rows x 8 groups of 4 rank columns
    const int r16 = threadIdx.x >> 3, c = (threadIdx.x & 7) * 4;
    const long m = rowblock * 16 + r16;
    if (m < M) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int p0 = 0; p0 < ks; p0 += 8) {   // eight pieces in flight (each load is a round trip to memory), added in piece order
        float u[8][4];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const float* src = part + ((long)min(p0 + q, ks - 1) * M + m) * 32 + c;
          if (coherent) {
#pragma unroll
            for (int e = 0; e < 4; ++e) u[q][e] = __hip_atomic_load(src + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          } else {
            const float4 w = *reinterpret_cast<const float4*>(src);
            u[q][0] = w.x, u[q][1] = w.y, u[q][2] = w.z, u[q][3] = w.w;
          }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q)
          if (p0 + q < ks) v.x += u[q][0], v.y += u[q][1], v.z += u[q][2], v.w += u[q][3];
      }
      const uint2 t = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
      *reinterpret_cast<uint2*>(T + m * 32 + c) = t;
      const uint2 sv = *reinterpret_cast<const uint2*>(S + (m / rps) * 32 + c);
      *reinterpret_cast<uint2*>(Ts + m * 32 + c) =
          make_uint2(pack_bf16x2(bf16lo(t.x) * bf16lo(sv.x), bf16hi(t.x) * bf16hi(sv.x)),
                     pack_bf16x2(bf16lo(t.y) * bf16lo(sv.y), bf16hi(t.y) * bf16hi(sv.y)));
    }
  }
}

__global__ __launch_bounds__(128) void lora_down_splitk_finalize_kernel(const float* __restrict__ part, int ks, long M,
                                                                        const bf16_t* __restrict__ S, int rps, bf16_t* __restrict__ T,
                                                                        bf16_t* __restrict__ Ts) {
  splitk_sum_rows(part, ks, M, blockIdx.x, S, rps, T, Ts, false);
}

template <bool TICKET>
__global__ __launch_bounds__(256) void lora_down_splitk_kernel(const bf16_t* __restrict__ X, long ldx, long M, int K, int kchunk,
                                                               const bf16_t* __restrict__ A, const bf16_t* __restrict__ S,
                                                               int rps, bf16_t* __restrict__ T, bf16_t* __restrict__ Ts,
                                                               float* __restrict__ part, int* __restrict__ counters) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
  static_assert(!TICKET, "the fence-free ticket form relies on the sc1 write-through / read-around path of gfx942 / gfx950");
#endif
  constexpr int RF = 2;
  __shared__ f32x4_t red[4][RF][64];
  __shared__ int s_last;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long row = (long)blockIdx.x * 16 + (lane & 15);
  const int g = lane >> 4;
  const bool ok = row < M;
  const int ks = gridDim.y, piece = blockIdx.y;
  const int k0 = piece * kchunk, k1 = min(K, k0 + kchunk);
  const __amdgpu_buffer_rsrc_t rx = aqlgemm::make_rsrc(X), ra = aqlgemm::make_rsrc(A);
  const uint32_t xoff = ok ? (uint32_t)(row * ldx + k0 + g * 8) * 2u : aqlgemm::OOB_ROW;
  const uint32_t aoff = (uint32_t)((lane & 15) * K + k0 + g * 8) * 2u;
  f32x4_t acc[RF];
#pragma unroll
  for (int f = 0; f < RF; ++f) acc[f] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  constexpr int D = 8;
  const int nsteps = (k1 - k0) / 32;   // wave-uniform (kchunk, K multiples of 32)
  for (int s0 = wave; s0 < nsteps; s0 += 4 * D) {
    aqlgemm::u32x4_t xv[D], av[D][RF];
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int s = s0 + 4 * d;
      if (s < nsteps) {   // scalar branch (see lora_down_skinny_body: MFMA ignores EXEC)
        const uint32_t so = (uint32_t)s * 64u;
        xv[d] = __builtin_amdgcn_raw_buffer_load_b128(rx, xoff + so, 0, 0);
#pragma unroll
        for (int f = 0; f < RF; ++f)
          av[d][f] = __builtin_amdgcn_raw_buffer_load_b128(ra, aoff + (uint32_t)(f * 16 * K) * 2u + so, 0, 0);
      }
    }
#pragma unroll
    for (int d = 0; d < D; ++d) {
      if (s0 + 4 * d < nsteps) {
#pragma unroll
        for (int f = 0; f < RF; ++f)
          acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8_t*>(&av[d][f]),
                                                           *reinterpret_cast<const bf16x8_t*>(&xv[d]), acc[f], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int f = 0; f < RF; ++f) red[wave][f][lane] = acc[f];
  __syncthreads();
  // wavefronts 0 / 1 sum fragment 0 / 1 over the four wavefronts: rows j = lane & 15, rank columns f*16 + g*4 + e
  if (wave < RF) {
    f32x4_t v = red[0][wave][lane];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const f32x4_t u = red[w][wave][lane];
      v[0] += u[0]; v[1] += u[1]; v[2] += u[2]; v[3] += u[3];
    }
    if (ok) {
      float* dst = part + ((long)piece * M + row) * 32 + wave * 16 + g * 4;
      if constexpr (TICKET) {   // device-scope (sc1, write-through) stores: the piece is in memory, not dirty in this XCD's L2
#pragma unroll
        for (int e = 0; e < 4; ++e) __hip_atomic_store(dst + e, v[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
  }
  if constexpr (!TICKET) return;   // the finalize launch adds the pieces
  // No fence (see the comment above the kernel): the pieces are the only data exchanged, they are written through (above) and
  // read around the L2 (splitk_sum_rows, coherent); the wait makes every wavefront's stores complete before the ticket is taken.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const int old = __hip_atomic_fetch_add(counters + blockIdx.x, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = old == ks - 1;
    if (s_last) __hip_atomic_store(counters + blockIdx.x, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // every other piece has taken its ticket
  }
  __syncthreads();
  if (!s_last) return;
  splitk_sum_rows(part, ks, M, blockIdx.x, S, rps, T, Ts, true);
}

// aql_lora_down for rank 32 with the K range split over workgroups when the row count alone cannot fill the chip; `part`
// (>= ks * M * 32 floats, ks <= 32) and `counters` (>= ceil(M / 16) ints, zero, left zero) are caller-owned scratch.  Falls
// through to aql_lora_down when no split pays (many rows or shallow K) or the scratch is too small.
extern "C" int aql_lora_down_splitk(const bf16_t* X, long ldx, long M, int K, const bf16_t* Adown, int r, const bf16_t* S,
                                    int rows_per_sample, bf16_t* T, bf16_t* Ts, float* part, size_t part_bytes, int* counters,
                                    size_t counters_bytes, hipStream_t stream) {
  AQL_CHECK_ARG(X && Adown && S && T && Ts, "aql_lora_down_splitk: null operand");
  AQL_CHECK_ARG(M > 0 && r > 0 && K > 0 && K % 8 == 0 && ldx % 8 == 0 && rows_per_sample > 0,
                "aql_lora_down_splitk: bad shape M=%ld r=%d K=%d", M, r, K);
  const long rb = (M + 15) / 16;
  int ks = 1;
  if (r == 32 && K % 32 == 0 && part != nullptr && counters != nullptr && rb * sizeof(int) <= counters_bytes) {
    static const int target = AQL_TUNE_INT("AQL_DOWN_SPLIT_WGS", 512);   // tuning hook: workgroups aimed at
    ks = (int)((target + rb - 1) / rb);
    if (ks > K / 256) ks = K / 256;          // at least 256 columns (two K steps per wavefront) per piece
    if (ks > 32) ks = 32;
    while (ks > 1 && (size_t)ks * (size_t)M * 32u * sizeof(float) > part_bytes) --ks;
  }
  if (ks <= 1) return aql_lora_down(X, ldx, M, K, Adown, r, S, rows_per_sample, T, Ts, nullptr, nullptr, stream);
  int kchunk = ((K / 32 + ks - 1) / ks) * 32;
  ks = (K + kchunk - 1) / kchunk;            // no empty pieces
  const char* tk = getenv("AQL_DOWN_TICKET");   // tuning hook, read per call (tests flip it): one launch, last arrival adds
  const bool ticket = tk != nullptr && tk[0] == '1';
  if (ticket) {
    hipLaunchKernelGGL(lora_down_splitk_kernel<true>, dim3((unsigned)rb, (unsigned)ks), dim3(256), 0, stream, X, ldx, M, K, kchunk,
                       Adown, S, rows_per_sample, T, Ts, part, counters);
  } else {
    hipLaunchKernelGGL(lora_down_splitk_kernel<false>, dim3((unsigned)rb, (unsigned)ks), dim3(256), 0, stream, X, ldx, M, K, kchunk,
                       Adown, S, rows_per_sample, T, Ts, part, counters);
    hipLaunchKernelGGL(lora_down_splitk_finalize_kernel, dim3((unsigned)rb), dim3(128), 0, stream, part, ks, M, S, rows_per_sample,
                       T, Ts);
  }
  AQL_CHECK_LAUNCH("aql_lora_down_splitk");
  return AQL_OK;
}

// n <= 32 rank-32 down products sharing M, S and rows_per_sample: T[i] = X[i].A[i]^T, Ts[i] = T[i] * S[m / rps];
// X[i] [M][K[i]] dense, A[i] [32][K[i]], K[i] % 32 == 0; T, Ts [n][M][32].  X, A, K are HOST arrays.
extern "C" int aql_lora_down_grouped(int n, const bf16_t* const* X, const bf16_t* const* A, const int* K, long M,
                                     const bf16_t* S, int rows_per_sample, bf16_t* T, bf16_t* Ts, hipStream_t stream) {
  AQL_CHECK_ARG(n >= 1 && n <= 32 && X && A && K && S && T && Ts && M > 0 && rows_per_sample > 0,
                "aql_lora_down_grouped: bad args");
  DownGroup g{};
  for (int i = 0; i < n; ++i) {
    AQL_CHECK_ARG(X[i] && A[i] && K[i] > 0 && K[i] % 32 == 0, "aql_lora_down_grouped: bad problem %d", i);
    g.X[i] = X[i], g.A[i] = A[i], g.K[i] = K[i];
  }
  hipLaunchKernelGGL(lora_down_skinny_grouped_kernel, dim3((unsigned)((M + 15) / 16), n), dim3(256), 0, stream, g, M, S,
                     rows_per_sample, T, Ts);
  AQL_CHECK_LAUNCH("aql_lora_down_grouped");
  return AQL_OK;
}

extern "C" int aql_conv3x3_fwd_pad(const bf16_t* X, int B, int Hin, int Win, int Cin, const bf16_t* Wk, const bf16_t* bias,
                                   int Cout, int stride, int upsample, int pad_lo, const bf16_t* rowbias, long rowbias_ld,
                                   const bf16_t* residual, bf16_t* Y, float* ws, size_t ws_bytes, hipStream_t stream);

extern "C" int aql_conv3x3_fwd(const bf16_t* X, int B, int Hin, int Win, int Cin, const bf16_t* Wk, const bf16_t* bias,
                               int Cout, int stride, int upsample, const bf16_t* rowbias, long rowbias_ld,
                               const bf16_t* residual, bf16_t* Y, float* ws, size_t ws_bytes, hipStream_t stream) {
  return aql_conv3x3_fwd_pad(X, B, Hin, Win, Cin, Wk, bias, Cout, stride, upsample, 1, rowbias, rowbias_ld, residual, Y, ws,
                             ws_bytes, stream);
}

// pad_lo = 1: torch padding=1.  pad_lo = 0 (stride 2 only): zero padding on the bottom/right edge only, i.e.
// F.pad(x, (0,1,0,1)) + Conv2d(padding=0) -- the Downsample2D of diffusers' AutoencoderKL encoder.
static int conv3x3_fwd_impl(const bf16_t* X, int B, int Hin, int Win, int Cin, const bf16_t* Wk, const bf16_t* bias,
                            int Cout, int stride, int upsample, int pad_lo, const bf16_t* rowbias, long rowbias_ld,
                            const bf16_t* residual, bf16_t* Y, float* ws, size_t ws_bytes, int* defer_splits, hipStream_t stream);

extern "C" int aql_conv3x3_fwd_pad(const bf16_t* X, int B, int Hin, int Win, int Cin, const bf16_t* Wk, const bf16_t* bias,
                                   int Cout, int stride, int upsample, int pad_lo, const bf16_t* rowbias, long rowbias_ld,
                                   const bf16_t* residual, bf16_t* Y, float* ws, size_t ws_bytes, hipStream_t stream) {
  return conv3x3_fwd_impl(X, B, Hin, Win, Cin, Wk, bias, Cout, stride, upsample, pad_lo, rowbias, rowbias_ld, residual, Y, ws, ws_bytes,
                          nullptr, stream);
}

// aql_conv3x3_fwd whose split-K finalize launch is left to the consumer (round 6): *splits_out = 1: Y is complete;
// *splits_out = s > 1: ws holds s fp32 partial slabs [s][B*Hout*Wout][Cout] and Y is NOT written -- finish with
// aql_groupnorm_silu_fwd_slabs (the GroupNorm behind the convolution, ResnetBlock2D.forward, original_unet.py:423-453) or aql_splitk_finalize.
extern "C" int aql_conv3x3_fwd_defer(const bf16_t* X, int B, int Hin, int Win, int Cin, const bf16_t* Wk, const bf16_t* bias,
                                     int Cout, int stride, int upsample, const bf16_t* rowbias, long rowbias_ld,
                                     const bf16_t* residual, bf16_t* Y, float* ws, size_t ws_bytes, int* splits_out,
                                     hipStream_t stream) {
  AQL_CHECK_ARG(splits_out != nullptr, "aql_conv3x3_fwd_defer: splits_out is null");
  *splits_out = 1;
  return conv3x3_fwd_impl(X, B, Hin, Win, Cin, Wk, bias, Cout, stride, upsample, 1, rowbias, rowbias_ld, residual, Y, ws, ws_bytes,
                          splits_out, stream);
}

// The finalize launch of a split-K GEMM on its own: C = bf16(sum_z slabs[z] + bias) + rowbias[m / rows_per_sample] + residual.
extern "C" int aql_splitk_finalize(const float* slabs, int splits, long M, int N, const bf16_t* bias, const bf16_t* rowbias,
                                   long rowbias_ld, int rows_per_sample, const bf16_t* residual, long ldr, bf16_t* C, long ldc,
                                   hipStream_t stream) {
  AQL_CHECK_ARG(slabs && C && splits >= 1 && M > 0 && N > 0 && N % 4 == 0 && ldc >= N, "aql_splitk_finalize: bad arguments");
  const long nchunk = M * (N / 4);
  int blocks = (int)((nchunk + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(splitk_finalize_kernel, dim3(blocks), dim3(256), 0, stream, slabs, splits, M, N, bias, rowbias,
                     rowbias_ld > 0 ? rowbias_ld : (long)N, rows_per_sample > 0 ? rows_per_sample : 1, residual, ldr, C, ldc);
  AQL_CHECK_LAUNCH("aql_splitk_finalize");
  return AQL_OK;
}

static int conv3x3_fwd_impl(const bf16_t* X, int B, int Hin, int Win, int Cin, const bf16_t* Wk, const bf16_t* bias,
                            int Cout, int stride, int upsample, int pad_lo, const bf16_t* rowbias, long rowbias_ld,
                            const bf16_t* residual, bf16_t* Y, float* ws, size_t ws_bytes, int* defer_splits, hipStream_t stream) {
  AQL_CHECK_ARG(X && Wk && Y, "aql_conv3x3_fwd: null operand");
  AQL_CHECK_ARG(pad_lo == 1 || (pad_lo == 0 && stride == 2 && Hin % 2 == 0 && Win % 2 == 0),
                "aql_conv3x3_fwd: pad_lo=0 needs stride 2 and even H, W");
  AQL_CHECK_ARG(Cin % 8 == 0 && Cout % 8 == 0, "aql_conv3x3_fwd: Cin/Cout must be multiples of 8 (%d,%d)", Cin, Cout);
  AQL_CHECK_ARG((stride == 1 || stride == 2) && (upsample == 0 || upsample == 1) && !(upsample && stride == 2),
                "aql_conv3x3_fwd: bad stride/upsample");
  ConvFwdLoader l;
  l.base = X;
  l.B = B;
  l.Hin = Hin;
  l.Win = Win;
  l.Cin = Cin;
  l.stride = stride;
  l.ups = upsample;
  l.pad = pad_lo;
  const int Hl = Hin << upsample, Wl = Win << upsample;
  l.Hout = (Hl + 2 - 3) / stride + 1;   // == Hl / 2 for pad (0,1) on even sizes as well
  l.Wout = (Wl + 2 - 3) / stride + 1;
  l.rows = B * l.Hout * l.Wout;
  l.K = 9 * Cin;
  GemmArgs<ConvFwdLoader, PlainLoader> g;
  g.a0 = l;
  g.a1 = l;
  g.b0 = plain(Wk, 9L * Cin, Cout, 9 * Cin);
  g.b1 = g.b0;
  g.ktiles0 = aql_cdiv(9 * Cin, BK);
  g.ktiles1 = 0;
  g.M = l.rows;
  g.N = Cout;
  OutSpec o{bias, rowbias, rowbias_ld, l.Hout * l.Wout, residual, Cout, Y, Cout, nullptr, 0, nullptr};
  o.defer_splits = defer_splits;
  return run_bf16_gemm(g, o, ws, ws_bytes, stream, "aql_conv3x3_fwd");
}

// dX[b,hi,wi,ci] = sum_{kh,kw,co} dY[b,ho,wo,co] * Wt[ci][(kh*3+kw)*Cout+co],  hi = ho*stride + kh - 1.
static int conv3x3_bwd_data_impl(const bf16_t* dY, int B, int Hin, int Win, int Cin, const bf16_t* Wt, int Cout, int stride, bf16_t* dX,
                                 float* ws, size_t ws_bytes, int* defer_splits, hipStream_t stream);

extern "C" int aql_conv3x3_bwd_data(const bf16_t* dY, int B, int Hin, int Win, int Cin, const bf16_t* Wt, int Cout,
                                    int stride, bf16_t* dX, float* ws, size_t ws_bytes, hipStream_t stream) {
  return conv3x3_bwd_data_impl(dY, B, Hin, Win, Cin, Wt, Cout, stride, dX, ws, ws_bytes, nullptr, stream);
}

// aql_conv3x3_bwd_data with the split-K finalize left to the consumer (see aql_conv3x3_fwd_defer): the GroupNorm backward in front of
// the convolution's input (aql_groupnorm_silu_bwd_slabs), or aql_splitk_finalize.
extern "C" int aql_conv3x3_bwd_data_defer(const bf16_t* dY, int B, int Hin, int Win, int Cin, const bf16_t* Wt, int Cout,
                                          int stride, bf16_t* dX, float* ws, size_t ws_bytes, int* splits_out, hipStream_t stream) {
  AQL_CHECK_ARG(splits_out != nullptr, "aql_conv3x3_bwd_data_defer: splits_out is null");
  *splits_out = 1;
  return conv3x3_bwd_data_impl(dY, B, Hin, Win, Cin, Wt, Cout, stride, dX, ws, ws_bytes, splits_out, stream);
}

static int conv3x3_bwd_data_impl(const bf16_t* dY, int B, int Hin, int Win, int Cin, const bf16_t* Wt, int Cout, int stride, bf16_t* dX,
                                 float* ws, size_t ws_bytes, int* defer_splits, hipStream_t stream) {
  AQL_CHECK_ARG(dY && Wt && dX, "aql_conv3x3_bwd_data: null operand");
  AQL_CHECK_ARG(Cin % 8 == 0 && Cout % 8 == 0 && (stride == 1 || stride == 2), "aql_conv3x3_bwd_data: bad shape");
  ConvBwdLoader l;
  l.base = dY;
  l.B = B;
  l.Hin = Hin;
  l.Win = Win;
  l.Cout = Cout;
  l.stride = stride;
  l.Hout = (Hin + 2 - 3) / stride + 1;
  l.Wout = (Win + 2 - 3) / stride + 1;
  l.rows = B * Hin * Win;
  l.K = 9 * Cout;
  GemmArgs<ConvBwdLoader, PlainLoader> g;
  g.a0 = l;
  g.a1 = l;
  g.b0 = plain(Wt, 9L * Cout, Cin, 9 * Cout);
  g.b1 = g.b0;
  g.ktiles0 = aql_cdiv(9 * Cout, BK);
  g.ktiles1 = 0;
  g.M = l.rows;
  g.N = Cin;
  OutSpec o{nullptr, nullptr, 0, 1, nullptr, 0, dX, Cin, nullptr, 0, nullptr};
  o.defer_splits = defer_splits;
  return run_bf16_gemm(g, o, ws, ws_bytes, stream, "aql_conv3x3_bwd_data");
}

// C[P,Q] (fp32) += alpha * sum_m U[m,P] * V[m,Q]   -- token-reduction GEMM for the LoRA weight gradients.
// The reduction over m is split across workgroups and combined with fp32 atomics, so C must hold the value to
// accumulate onto (zero for a fresh gradient).
extern "C" int aql_gemm_tn_f32(const bf16_t* U, long ldu, const bf16_t* V, long ldv, long M, int P, int Q, float alpha,
                               float* C, long ldc, hipStream_t stream) {
  AQL_CHECK_ARG(U && V && C, "aql_gemm_tn_f32: null operand");
  AQL_CHECK_ARG(M > 0 && P % 8 == 0 && Q % 8 == 0 && ldu % 8 == 0 && ldv % 8 == 0 && M < (1L << 31),
                "aql_gemm_tn_f32: bad shape M=%ld P=%d Q=%d", M, P, Q);
  // the narrow side (the LoRA rank) always goes on the weight-side (BN) operand; swap + transposed write if needed
  const bool swap = (P <= 32 && Q > 32);
  GemmArgs<TransLoader, TransLoader> g;
  g.a0.base = swap ? V : U;
  g.a0.ld = swap ? ldv : ldu;
  g.a0.rows = swap ? Q : P;
  g.a0.K = (int)M;
  g.b0.base = swap ? U : V;
  g.b0.ld = swap ? ldu : ldv;
  g.b0.rows = swap ? P : Q;
  g.b0.K = (int)M;
  g.a1 = g.a0;
  g.b1 = g.b0;
  g.ktiles0 = aql_cdiv(M, BK);
  g.ktiles1 = 0;
  g.M = g.a0.rows;
  g.N = g.b0.rows;
  g.epi = EpiParams{};
  g.epi.Cf = C;
  g.epi.ldcf = ldc;
  g.epi.alpha = alpha;
  g.epi.trans_out = swap ? 1 : 0;
  g.m_fast = 0;
  int tiles = 0;
  int cfg = pick_cfg(g.M, g.N, g.ktiles0, false, &tiles);
  if (cfg == 3) cfg = 2, tiles = aql_cdiv(g.M, 128) * aql_cdiv(g.N, 64);
  int splits = (384 + tiles - 1) / tiles;  // measured on MI355X (tools/probe_tn.py): 300-400 workgroups
  if (splits > g.ktiles0 / 4) splits = g.ktiles0 / 4;
  if (splits > 48) splits = 48;
  if (splits < 1) splits = 1;
#ifdef AQL_EXPERIMENTS
  if (const char* e = getenv("AQL_TN_SPLITS")) splits = atoi(e) < g.ktiles0 ? atoi(e) : g.ktiles0;  // tuning hook
#endif
  g.splits = splits;
  launch_cfg<TransLoader, TransLoader, EPI_ATOMIC>(cfg, 1, g, stream);
  AQL_CHECK_LAUNCH("aql_gemm_tn_f32");
  return AQL_OK;
}

// dst [cols][rows] = src [rows][cols]^T (bf16); rows % 8 == 0 and cols % 8 == 0
extern "C" int aql_transpose_bf16(const bf16_t* src, long rows, int cols, long ld, bf16_t* dst, hipStream_t stream) {
  AQL_CHECK_ARG(src && dst && rows > 0 && cols > 0 && rows % 8 == 0 && cols % 8 == 0 && ld % 8 == 0,
                "aql_transpose_bf16: bad shape rows=%ld cols=%d", rows, cols);
  hipLaunchKernelGGL(transpose_bf16_kernel, dim3((unsigned)((rows + 63) / 64), (cols + 63) / 64), dim3(256), 0, stream, src,
                     rows, cols, ld, dst);
  AQL_CHECK_LAUNCH("aql_transpose_bf16");
  return AQL_OK;
}

// C[M,N] (fp32) += alpha * A[M,K] . B[N,K]^T   (bf16 operands, K contiguous): the wide-rank (r > 32) LoRA weight
// gradients dA = dT^T.X and dB = dY^T.Ts after both operands were transposed once -- the pipelined NT kernels run
// 5-7x faster than the transposing-loader kernel on these shapes.  ws holds the split-K slabs (>= M*N*4 bytes).
extern "C" int aql_gemm_nt_f32_accum(const bf16_t* A, long lda, const bf16_t* B, long ldb, long M, int N, long K,
                                     float alpha, float* C, long ldc, float* ws, size_t ws_bytes, hipStream_t stream) {
  AQL_CHECK_ARG(A && B && C && ws, "aql_gemm_nt_f32_accum: null operand");
  AQL_CHECK_ARG(M > 0 && N % 8 == 0 && K % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && K < (1L << 31) &&
                    (size_t)M * N * 4u <= ws_bytes,
                "aql_gemm_nt_f32_accum: bad shape M=%ld N=%d K=%ld", M, N, K);
  GemmArgs<PlainLoader, PlainLoader> g;
  g.a0 = plain(A, lda, M, (int)K);
  g.b0 = plain(B, ldb, N, (int)K);
  g.a1 = g.a0;
  g.b1 = g.b0;
  g.ktiles0 = aql_cdiv(K, BK);
  g.ktiles1 = 0;
  g.M = (int)M;
  g.N = N;
  g.epi = EpiParams{};
  g.epi.rows_per_sample = 1;
  g.m_fast = 0;
  int tiles = 0, cfg = 0, pd = 1;
  pick_tile(g.M, g.N, g.ktiles0, true, &cfg, &tiles, &pd);
  int splits = 256 / (tiles > 0 ? tiles : 1);
  if (splits < 1) splits = 1;
  while (splits > 1 && (g.ktiles0 / splits < 8 || (size_t)splits * (size_t)M * (size_t)N * 4u > ws_bytes)) --splits;
  g.splits = splits;
  if (pd == 0) pd = (tiles * splits <= 288) ? 13 : 12;
  g.epi.Cf = ws;
  g.epi.ldcf = N;
  launch_cfg<PlainLoader, PlainLoader, EPI_SLAB>(cfg, pd, g, stream);
  AQL_CHECK_LAUNCH("aql_gemm_nt_f32_accum");
  const long nchunk = M * (N / 4);
  int blocks = (int)((nchunk + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(splitk_accum_kernel, dim3(blocks), dim3(256), 0, stream, ws, splits, M, N, alpha, C, ldc);
  AQL_CHECK_LAUNCH("aql_gemm_nt_f32_accum");
  return AQL_OK;
}

// ---- grouped weight-gradient GEMMs ---------------------------------------------------------------------------
// aql_tn_desc_fill writes one host-side descriptor (64 bytes) for  C[P,Q] += alpha * U[M,P]^T V[M,Q]  with
// min(P,Q) <= 32 and returns the number of workgroups it needs (0 if the shape is not groupable);
// aql_gemm_tn_grouped launches all of them at once from a DEVICE copy of the table.
extern "C" int aql_tn_desc_fill(void* host_desc, const bf16_t* U, long ldu, const bf16_t* V, long ldv, long M, int P,
                                int Q, float alpha, float* C, long ldc, int first_block) {
  if (host_desc == nullptr || U == nullptr || V == nullptr || C == nullptr) return 0;
  if (P % 8 || Q % 8 || ldu % 8 || ldv % 8 || M <= 0 || M >= (1L << 31)) return 0;
  const bool swap = (P <= 32 && Q > 32);
  const int a_rows = swap ? Q : P, b_rows = swap ? P : Q;
  if (b_rows > 32) return 0;
  TnGroupDesc d;
  d.a = swap ? V : U;
  d.b = swap ? U : V;
  d.lda = swap ? ldv : ldu;
  d.ldb = swap ? ldu : ldv;
  d.C = C;
  d.ldc = ldc;
  d.M = (int)M;
  d.a_rows = a_rows;
  d.b_rows = b_rows;
  const int tiles = aql_cdiv(a_rows, 128);
  const int ktiles = aql_cdiv(M, BK);
  int splits = (384 + tiles - 1) / tiles;
  if (splits > ktiles / 4) splits = ktiles / 4;
  if (splits > 48) splits = 48;
  if (splits < 1) splits = 1;
  d.splits = splits;
  d.first_block = first_block;
  d.trans_out = swap ? 1 : 0;
  d.alpha = alpha;
  d.pad = 0;
  memcpy(host_desc, &d, sizeof(d));
  return tiles * splits;
}

extern "C" int aql_gemm_tn_grouped(const void* dev_descs, int n, int total_blocks, hipStream_t stream) {
  AQL_CHECK_ARG(dev_descs && n > 0 && total_blocks > 0, "aql_gemm_tn_grouped: bad args");
  static_assert(sizeof(TnGroupDesc) == 80, "descriptor layout is part of the ABI");
  hipLaunchKernelGGL(gemm_tn_grouped_kernel, dim3(total_blocks), dim3(NTHREADS), 0, stream,
                     static_cast<const TnGroupDesc*>(dev_descs), n, 0);
  AQL_CHECK_LAUNCH("aql_gemm_tn_grouped");
  return AQL_OK;
}

// Sub-range form: descriptors [first, first + n) of the same table, i.e. one gradient bucket.  `block_base` is the
// first_block of descriptor `first`, `n_blocks` the workgroups of the range.
extern "C" int aql_gemm_tn_grouped_range(const void* dev_descs, int first, int n, int block_base, int n_blocks,
                                         hipStream_t stream) {
  AQL_CHECK_ARG(dev_descs && first >= 0 && n > 0 && block_base >= 0 && n_blocks > 0,
                "aql_gemm_tn_grouped_range: bad args");
  hipLaunchKernelGGL(gemm_tn_grouped_kernel, dim3(n_blocks), dim3(NTHREADS), 0, stream,
                     static_cast<const TnGroupDesc*>(dev_descs) + first, n, block_base);
  AQL_CHECK_LAUNCH("aql_gemm_tn_grouped_range");
  return AQL_OK;
}
