// X-stationary ("strip") form of the one-launch rank-32 LoRA linear for SHORT K (utils/lora_modules.py:9-26 + 56-62):
//     Y = X.W^T + ((X.A^T) * S).Bup^T + bias + residual          (optionally with the GEGLU epilogue)
//
// Why: with K = 320 / 640 a 128x160 output tile has a 5-10 tile K loop, and an in-kernel timeline of lora_gemm_kernel
// (tools/trace_lora.py, MI355X) shows a workgroup spending ~60 % of its life OUTSIDE that loop -- kernel-argument loads, ring
// fill, the T -> Ts exchange through LDS, the C tile round trip through LDS, the store tail -- while every one of the N/160 column
// tiles of a row block re-reads the same X rows from L2 and recomputes the same T = X.A^T (+20 % MFMAs).  Here ONE workgroup
// owns a strip of BM rows for ALL N columns:
//   * the strip's X panel [BM][K] is fetched once and stays in LDS (80 KB), T = X.A^T is computed once, Ts = bf16(T) * S stays
//     in LDS (and is written out with T for the backward pass);
//   * the weights stream through an LDS-DMA ring as one uniform sequence of 160-row x 64-column tiles: K/64 tiles of the LoRA
//     down matrix A, then per column tile K/64 tiles of W followed by ONE tile holding the Bup panel (32 rank columns), which
//     is multiplied with Ts instead of X -- the LoRA up-projection is just a (K/64 + 1)-th k-tile of every column tile;
//   * wavefronts 4-7 only issue the DMAs (as in lora_gemm_kernel_w), wavefronts 0-3 only read fragments and issue MFMAs; a compute
//     wavefront owns BM/4 rows x all 160 columns of the tile, so the [80 value | 80 gate] halves of a GEGLU tile, the bias, the
//     residual and the output rows all live in ONE wavefront's accumulators: the epilogue runs from registers, without LDS, without
//     a workgroup barrier, while the loader wavefronts already stream the next column tile's weights.
// Arithmetic (accumulation order, rounding points) is that of lora_gemm_kernel: results are bit-identical.
#pragma once
#include "aql_gemm.cuh"

namespace aqlstrip {
using namespace aqlgemm;

constexpr int SBN = 160;          // columns per tile (GEGLU: 80 value + 80 gate)
constexpr int SLR = 32;           // LoRA rank
constexpr int RING_TILE = SBN * 128;

struct StripArgs {
  PlainLoader x;        // activations [M][K]
  PlainLoader w;        // weights [N][K] (GEGLU: gsplit / goff select the value / gate rows of a tile)
  PlainLoader a;        // LoRA down [32][K]
  PlainLoader bup;      // LoRA up [N][32] as a one-tile operand (same row mapping as w)
  const bf16_t* S;      // [nsamples][32]
  const bf16_t* bias;
  const bf16_t* residual;
  long ldr;
  bf16_t* Y;            // [M][ldy]   plain: the output; GEGLU: the pre-activation H [M][2F] or null
  long ldy;
  bf16_t* G;            // GEGLU: activated output [M][ldg]
  long ldg;
  bf16_t *T, *Ts;       // [M][32]
  int M, N, rps;
  int row0;             // rows below row0 have no LoRA term (clean half of a twin batch); multiple of BM or 0
  int geglu_F;          // F > 0: GEGLU tiles
  int c_row0;           // GEGLU: H is written for rows >= c_row0 only
  long long* trace;     // -DAQL_TRACE_L builds: per-workgroup phase timestamps (tools/trace_strip.py), else null
};

#ifdef AQL_TRACE_L
#define STRACE(slot) do { if (str) str[slot] = __builtin_readcyclecounter(); } while (0)
#else
#define STRACE(slot) do { } while (0)
#endif

template <int BM, int KT, int NSTG>
__global__ __launch_bounds__(2 * NTHREADS) void lora_strip_kernel(const StripArgs a) {
  constexpr int WM = BM / 4, FM = WM / 16, FN = SBN / 16;
  static_assert(FM >= 1 && BM % 64 == 0, "a compute wavefront owns BM/4 >= 16 rows");
  constexpr int XT = BM * 128;                       // bytes of one 64-column tile of the X panel
  constexpr int PANEL = KT * XT, TS_BYTES = XT;      // Ts uses the A-tile layout (k-step 0 of a tile)
  constexpr int LDS_BYTES = PANEL + TS_BYTES + NSTG * RING_TILE;
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
  __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
  char* const panel = lds;
  char* const sTs = lds + PANEL;
  char* const ring = lds + PANEL + TS_BYTES;
  constexpr int NLD = SBN / 32;                      // DMAs per loader thread per ring tile

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wave >= 4;
  const int ltid = tid & 255;
  const int m0 = blockIdx.x * BM;
  const bool lora_on = m0 + BM > a.row0;             // block-uniform
  const int gF = a.geglu_F;
  const int tile_cols = gF ? SBN / 2 : SBN;          // output columns (per half) a tile advances
  const int ntn = (gF ? gF : a.N) / tile_cols;
  const int R = (lora_on ? KT : 0) + ntn * (KT + (lora_on ? 1 : 0));   // ring tiles of this strip

  if (loader) {
    // ---- X panel: KT tiles, once
    DmaStager<BM, PlainLoader> sx;
    sx.begin(a.x, a.x, false, m0, ltid, 0, KT, KT);
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) sx.dma(panel + kt * XT, wave - 4);
    // ---- ring: one uniform tile sequence (see header)
    DmaStager<SBN, PlainLoader> sb;
    int gen = 0, ph_left = 0, next_n = lora_on ? -1 : 0, cur_n = 0;
    bool bup_pending = false;
    auto issue_next = [&](int stage) {
      if (gen < R && ph_left == 0) {   // uniform: enter the next phase of the sequence
        if (next_n < 0) {
          sb.begin(a.a, a.a, false, 0, ltid, 0, KT, KT);
          ph_left = KT;
          next_n = 0;
        } else if (!bup_pending) {
          cur_n = next_n++;
          sb.begin(a.w, a.w, false, cur_n * tile_cols, ltid, 0, KT, KT);
          ph_left = KT;
          bup_pending = lora_on;
        } else {
          sb.begin(a.bup, a.bup, false, cur_n * tile_cols, ltid, 0, 1, 1);
          ph_left = 1;
          bup_pending = false;
        }
      }
      sb.dma(ring + stage * RING_TILE, wave - 4);   // past the end of a phase / of the strip: zero fill without traffic
      --ph_left;
      ++gen;
    };
    if (R == 0) return;
#pragma unroll
    for (int u = 0; u < NSTG - 1; ++u) issue_next(u);
    int wr = NSTG - 1;
    for (int t = 0; t < R; ++t) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTG - 2) * NLD) : "memory");   // tile t (and, at t = 0, the X panel) landed
      __builtin_amdgcn_s_barrier();    // B(t): tile t is visible; nobody reads tile t-1 any more
      asm volatile("" ::: "memory");
      issue_next(wr);                  // refill the stage of tile t-1 with tile t+NSTG-1
      wr = (wr + 1 == NSTG) ? 0 : wr + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }

  // ------------------------------------------------------------------------------------------ compute wavefronts
  const int wm0 = wave * WM;
  const int arow = wm0 + (lane & 15), g4 = lane >> 4;
#ifdef AQL_TRACE_L
  long long* str = (a.trace != nullptr && tid == 0) ? a.trace + (long)blockIdx.x * 128 : nullptr;
#endif
  STRACE(0);
  int rd = 0;
  auto next_stage = [&]() {
    const char* s = ring + rd * RING_TILE;
    rd = (rd + 1 == NSTG) ? 0 : rd + 1;
    return s;
  };

  // ---- T = X.A^T (rows of this wavefront, all 32 rank columns), Ts = bf16(T) * S -> this wavefront's rows of the Ts tile
  if (lora_on) {
    uint2 srow[FM][2];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const long m = (long)m0 + wm0 + i * 16 + (lane & 15);
      const bool ok = m < a.M;
#pragma unroll
      for (int t = 0; t < 2; ++t)
        srow[i][t] = epi_mask2(*reinterpret_cast<const uint2*>(a.S + (ok ? (long)((uint32_t)m / (uint32_t)a.rps) * SLR + t * 16 + g4 * 4 : 0)), ok);
    }
    f32x4_t tacc[FM][2];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int t = 0; t < 2; ++t) tacc[i][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < KT; ++kt) {
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const char* sB = next_stage();
      const char* sA = panel + kt * XT;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int chunk = ks * 4 + g4;
        bf16x8_t fa[FM], fl[2];
#pragma unroll
        for (int i = 0; i < FM; ++i) fa[i] = *reinterpret_cast<const bf16x8_t*>(sA + lds_off(arow + i * 16, chunk));
#pragma unroll
        for (int t = 0; t < 2; ++t) fl[t] = *reinterpret_cast<const bf16x8_t*>(sB + lds_off(t * 16 + (lane & 15), chunk));
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int t = 0; t < 2; ++t) tacc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fl[t], fa[i], tacc[i][t], 0, 0, 0);
      }
    }
    // tacc[i][t][e]: row m0 + wm0 + 16 i + (lane & 15), rank column 16 t + 4 (lane >> 4) + e
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int row = arow + i * 16;
      const long m = (long)m0 + row;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int r = t * 16 + g4 * 4;
        const uint2 tv = make_uint2(pack_bf16x2(tacc[i][t][0], tacc[i][t][1]), pack_bf16x2(tacc[i][t][2], tacc[i][t][3]));
        const uint2 sv = srow[i][t];
        const uint2 ts = make_uint2(pack_bf16x2(bf16lo(tv.x) * bf16lo(sv.x), bf16hi(tv.x) * bf16hi(sv.x)),
                                    pack_bf16x2(bf16lo(tv.y) * bf16lo(sv.y), bf16hi(tv.y) * bf16hi(sv.y)));
        *reinterpret_cast<uint2*>(sTs + lds_off(row, r >> 3) + (r & 7) * 2) = ts;
        if (m < a.M) {
          *reinterpret_cast<uint2*>(a.T + m * SLR + r) = tv;
          *reinterpret_cast<uint2*>(a.Ts + m * SLR + r) = ts;
        }
      }
    }
  }

  STRACE(1);
  // ---- column tiles
  for (int n = 0; n < ntn; ++n) {
    const int n0 = n * tile_cols;
#ifdef AQL_TRACE_L
    if (n < 30) STRACE(4 + 4 * n);
#endif
    // this lane's bias values and residual words of the tile: in flight during the tile's K loop
    uint2 biasr[FN];
    epi_load_bias<FN>(biasr, a.bias, a.w.base, n0, 0, lane, a.N, gF, SBN / 2);
    uint2 resr[FM][FN];
    const bool has_res = a.residual != nullptr;   // (never with GEGLU)
    if (has_res) {
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const long m = (long)m0 + arow + i * 16;
        const bool okm = m < a.M;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const int nn = n0 + j * 16 + g4 * 4;
          const bool ok = okm & (nn < a.N);
          resr[i][j] = *reinterpret_cast<const uint2*>(a.residual + (ok ? m * a.ldr + nn : 0));
        }
      }
    }
    f32x4_t acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < KT; ++kt) {
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const char* sB = next_stage();
      const char* sA = panel + kt * XT;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int chunk = ks * 4 + g4;
        bf16x8_t fa[FM], fb[FN];
#pragma unroll
        for (int i = 0; i < FM; ++i) fa[i] = *reinterpret_cast<const bf16x8_t*>(sA + lds_off(arow + i * 16, chunk));
#pragma unroll
        for (int j = 0; j < FN; ++j) fb[j] = *reinterpret_cast<const bf16x8_t*>(sB + lds_off(j * 16 + (lane & 15), chunk));
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
      }
    }
#ifdef AQL_TRACE_L
    if (n < 30) STRACE(5 + 4 * n);
#endif
    if (lora_on) {   // the Bup tile: one k-step against Ts
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const char* sB = next_stage();
      bf16x8_t fa[FM], fb[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) fa[i] = *reinterpret_cast<const bf16x8_t*>(sTs + lds_off(arow + i * 16, g4));
#pragma unroll
      for (int j = 0; j < FN; ++j) fb[j] = *reinterpret_cast<const bf16x8_t*>(sB + lds_off(j * 16 + (lane & 15), g4));
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
    }

#ifdef AQL_TRACE_L
    if (n < 30) STRACE(6 + 4 * n);
#endif
    // ---- epilogue from registers.  acc[i][j][e]: row m0 + wm0 + 16 i + (lane & 15), tile column 16 j + 4 (lane >> 4) + e
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const long m = (long)m0 + arow + i * 16;
      if (m >= a.M) continue;
      if (gF) {
#pragma unroll
        for (int j = 0; j < FN / 2; ++j) {
          const int nn = n0 + j * 16 + g4 * 4;
          if (nn >= gF) continue;
          const uint2 bv = biasr[j], bg = biasr[j + FN / 2];
          const f32x4_t cv = acc[i][j], cg = acc[i][j + FN / 2];
          const uint2 hv = make_uint2(pack_bf16x2(cv[0] + bf16lo(bv.x), cv[1] + bf16hi(bv.x)), pack_bf16x2(cv[2] + bf16lo(bv.y), cv[3] + bf16hi(bv.y)));
          const uint2 hg = make_uint2(pack_bf16x2(cg[0] + bf16lo(bg.x), cg[1] + bf16hi(bg.x)), pack_bf16x2(cg[2] + bf16lo(bg.y), cg[3] + bf16hi(bg.y)));
          if (a.Y != nullptr && m >= a.c_row0) {
            *reinterpret_cast<uint2*>(a.Y + m * a.ldy + nn) = hv;
            *reinterpret_cast<uint2*>(a.Y + m * a.ldy + gF + nn) = hg;
          }
          const uint2 o = make_uint2(pack_bf16x2(bf16lo(hv.x) * gelu_erf(bf16lo(hg.x)), bf16hi(hv.x) * gelu_erf(bf16hi(hg.x))),
                                     pack_bf16x2(bf16lo(hv.y) * gelu_erf(bf16lo(hg.y)), bf16hi(hv.y) * gelu_erf(bf16hi(hg.y))));
          *reinterpret_cast<uint2*>(a.G + m * a.ldg + nn) = o;
        }
      } else {
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const int nn = n0 + j * 16 + g4 * 4;
          if (nn >= a.N) continue;
          const uint2 bb = biasr[j];
          const f32x4_t c = acc[i][j];
          uint2 v = make_uint2(pack_bf16x2(c[0] + bf16lo(bb.x), c[1] + bf16hi(bb.x)), pack_bf16x2(c[2] + bf16lo(bb.y), c[3] + bf16hi(bb.y)));
          if (has_res) {
            const uint2 r = resr[i][j];
            v = make_uint2(pack_bf16x2(bf16lo(v.x) + bf16lo(r.x), bf16hi(v.x) + bf16hi(r.x)),
                           pack_bf16x2(bf16lo(v.y) + bf16lo(r.y), bf16hi(v.y) + bf16hi(r.y)));
          }
          *reinterpret_cast<uint2*>(a.Y + m * a.ldy + nn) = v;
        }
      }
    }
#ifdef AQL_TRACE_L
    if (n < 30) STRACE(7 + 4 * n);
#endif
  }
  STRACE(2);
}

template <int BM, int KT, int NSTG>
inline void launch_strip(const StripArgs& a, hipStream_t stream) {
  hipLaunchKernelGGL((lora_strip_kernel<BM, KT, NSTG>), dim3(aql_cdiv(a.M, BM)), dim3(2 * NTHREADS), 0, stream, a);
}

}  // namespace aqlstrip
