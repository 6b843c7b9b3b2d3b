// Fused rank-32 watermark-LoRA linear in ONE launch (utils/lora_modules.py:9-26 + 56-62):
//     T  = X.A^T                  (side accumulator, shares the X fragments of the main product)
//     Ts = T * S[sample]          (bf16, like the reference's `down(x) @ diag_embed(scale)` under autocast)
//     Y  = X.W^T + Ts.Bup^T + bias + residual
// and, with the operands exchanged, its backward-data form  dTs = dY.Bup, dT = dTs * S, dX = dY.W + dT.A.
//
// Why: the two-launch form (aql_lora_down, then aql_gemm_bf16 with Ts.Bup^T as a second K segment) spends 384 launches
// per step on the skinny T products -- each 5-9 us of ramp and drain for <= 10 MB of traffic, 3.5 ms of a 28.6 ms step by
// removal, the largest single item -- and reads X twice.  Here the workgroup that owns output tile (m, n) also stages the
// 32 rows of A for every K tile (4 KB per stage next to the 128x160 / 64x160 / ... operand tiles, one more LDS-DMA per
// thread) and accumulates T[m-tile, 32] with FM extra MFMAs per k-step (+20 % on a 160-wide tile) from the activation
// fragments it already holds.  After the K loop T is scaled, rounded to bf16, written into the idle stage-0 A tile in
// fragment layout together with the Bup[n-tile, 32] panel, and ONE more k-step adds Ts.Bup^T into the same accumulators.
// The n-tile-0 workgroups write T and Ts for the backward pass.  T is recomputed by every n-tile of a row block; that
// redundancy is the 20 %.
//
// Same LDS-DMA ring as gemm_body_d (aql_gemm.cuh): NSTG stages, one barrier per K tile, counted vmcnt waits.  No split-K
// (T must be complete before the up-projection): deep-K / small-grid shapes return AQL_NOT_FUSED and the caller uses the
// two-launch path.
#include <algorithm>
#include <type_traits>
#include "aql_gemm.cuh"
#include "aql_gemm_lora_kgroups.cuh"
#include "aql_gemm_lora_t256.cuh"
#include <stdlib.h>
#include <string.h>

using namespace aqlgemm;

#define AQL_NOT_FUSED 100

namespace {

constexpr int LR = 32;  // LoRA rank handled here

// -DAQL_TRACE_L: per-workgroup phase timestamps of the 4-wave kernel (tools/trace_lora.py); the buffer address comes from
// the environment (AQL_TRACE_BUF) and travels in EpiParams::Cf, which the bf16 epilogue does not use
#ifdef AQL_TRACE_L
#define LTRACE(slot) do { if (ltr) ltr[slot] = __builtin_readcyclecounter(); } while (0)
#else
#define LTRACE(slot) do { } while (0)
#endif

constexpr int MAXG = 32;  // LoRA linears per grouped launch

struct LoraParams {
  const bf16_t* S;    // [nsamples][32] bf16 scale rows
  const bf16_t* Bup;  // [N][32]
  bf16_t* T;          // [M][32] out (unscaled); grouped: [ngroups][M][32]
  bf16_t* Ts;         // [M][32] out (scaled);   grouped: [ngroups][M][32]
  int rps;            // rows per sample
  // Rows below row0 carry an all-zero scale row (the clean half of a twin batch, ppft_train.py:1026-1029): tiles that lie
  // entirely below it skip the LoRA branch -- no A-tile traffic, no T MFMAs, no up-projection k-step, no T / Ts output.
  int row0;
  // Grouped launch: the N output columns are the concatenation of `ngroups` independent LoRA linears that share X (q|k|v of a
  // self-attention; the k|v projections of the text states of all 16 cross-attentions).  Group i owns columns
  // [col_start[i], col_start[i+1]) (multiples of the tile width), rows [32 i, 32 i + 32) of the stacked A and slab i of T / Ts;
  // W, Bup and bias are stacked along N, so their row index stays the absolute column.  0 = one linear.
  int ngroups;
  int col_start[MAXG + 1];
};

template <int BM, int BN, int WM, int WN, int NSTG>
__global__ __launch_bounds__(NTHREADS) void lora_gemm_kernel(const GemmArgs<PlainLoader, PlainLoader> g, const PlainLoader la,
                                                             const LoraParams lp) {
  constexpr int FM = WM / 16, FN = WN / 16;
  constexpr int WAVES_N = BN / WN;
  static_assert((BM / WM) * (BN / WN) == 4, "4 wavefronts per workgroup");
  constexpr int FT = LR / (16 * WAVES_N);  // T fragments (16 rank columns each) per wavefront
  static_assert(FT >= 1, "at most two wavefronts along N");
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, L_BYTES = LR * 128;
  constexpr int STAGE = A_BYTES + B_BYTES + L_BYTES;
  constexpr int C_PITCH = (BN + 8) * 2;
  constexpr int LDS_BYTES = (NSTG * STAGE > BM * C_PITCH) ? NSTG * STAGE : BM * C_PITCH;
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
  __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];

  // XCD-aware remap (as gemm_kernel_d): every XCD gets a contiguous run of logical tiles
  const int nblk = gridDim.x, bid = blockIdx.x;
  const int qq = nblk >> 3, rr = nblk & 7, xcd = bid & 7;
  const int block_x = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef AQL_TRACE_L
  long long* ltr = (g.epi.Cf != nullptr && tid == 0) ? reinterpret_cast<long long*>(g.epi.Cf) + (long)bid * 16 : nullptr;
  LTRACE(0);
  if (ltr) {
    ltr[8] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));    // HW_ID
    ltr[9] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));   // XCC_ID
  }
#endif
  const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * WN;
  const int wt0 = (wave % WAVES_N) * FT;  // first T fragment of this wavefront
  const int tiles_n = (g.N + BN - 1) / BN, tiles_m = (g.M + BM - 1) / BM;
  int tile_m, tile_n;
  if (g.m_fast) {
    tile_n = block_x / tiles_m;
    tile_m = block_x - tile_n * tiles_m;
  } else {
    tile_m = block_x / tiles_n;
    tile_n = block_x - tile_m * tiles_n;
  }
  const int gF = g.epi.geglu_F;  // GEGLU tiles: [80 value | 80 gate] columns (aql_gemm.cuh)
  const int m0 = tile_m * BM, n0 = tile_n * (gF ? BN / 2 : BN);
  const int kt_end = g.ktiles0;
  int grp = 0;
  if (lp.ngroups > 0)
    while (grp + 1 < lp.ngroups && n0 >= lp.col_start[grp + 1]) ++grp;
  const bool t_writer = lp.ngroups > 0 ? (n0 == lp.col_start[grp]) : (tile_n == 0);
  bf16_t* const Tg = lp.T + (long)grp * g.M * LR;
  bf16_t* const Tsg = lp.Ts + (long)grp * g.M * LR;
  PlainLoader lag = la;
  lag.base += (long)grp * LR * la.ld;
  const bool lora_on = m0 + BM > lp.row0;   // block-uniform
  if (!lora_on) lag.rows = 0;              // every row out of range: the stage's A DMAs are zero fills without traffic

  DmaStager<BM, PlainLoader> sa;
  DmaStager<BN, PlainLoader> sb;
  DmaStager<LR, PlainLoader> sl;
  constexpr int NLD = BM / 32 + BN / 32 + 1;
  sa.begin(g.a0, g.a0, false, m0, tid, 0, kt_end, kt_end);
  sb.begin(g.b0, g.b0, false, n0, tid, 0, kt_end, kt_end);
  sl.begin(lag, lag, false, 0, tid, 0, kt_end, kt_end);
  LTRACE(10);

  f32x4_t acc[FM][FN], tacc[FM][FT];
#pragma unroll
  for (int i = 0; i < FM; ++i) {
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < FT; ++t) tacc[i][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }

  uint2 biasr[FN];  // this lane's bias values, fetched before the K loop (epi_load_bias, aql_gemm.cuh)
  epi_load_bias<FN>(biasr, g.epi.bias, g.b0.base, n0, wn0, lane, g.N, gF, BN / 2);
  // the up-projection panel Bup[n-tile, 32] and this lane's scale rows are fetched now, so that their latency hides under
  // the K loop instead of sitting between the loop and the final k-step
  constexpr int NBP = (BN * 4 + NTHREADS - 1) / NTHREADS;
  uint4 bup[NBP];
#pragma unroll
  for (int u = 0; u < NBP; ++u) {
    const int id = tid + u * NTHREADS, row = id >> 2, c = id & 3;
    const int brow = epi_bias_col(n0, row, gF, BN / 2);
    const bool ok = (id < BN * 4) & (brow < g.N) & lora_on;   // unconditional load, clamped address, AND-mask (see epi_load_bias)
    bup[u] = epi_mask4(*reinterpret_cast<const uint4*>(lp.Bup + (ok ? (long)brow * LR + c * 8 : 0)), ok);
  }
  uint2 srow[FM][FT];
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const long m = (long)m0 + wm0 + i * 16 + (lane & 15);
#pragma unroll
    for (int t = 0; t < FT; ++t) {
      const bool ok = (m < g.M) & lora_on;
      srow[i][t] = epi_mask2(*reinterpret_cast<const uint2*>(lp.S + (ok ? (long)((uint32_t)m / (uint32_t)lp.rps) * LR + (wt0 + t) * 16 + (lane >> 4) * 4 : 0)), ok);
    }
  }

  auto issue = [&](int stage) {
    char* sA = lds + stage * STAGE;
    sa.dma(sA, wave);
    sb.dma(sA + A_BYTES, wave);
    sl.dma(sA + A_BYTES + B_BYTES, wave);
  };
  LTRACE(11);
#pragma unroll
  for (int u = 0; u < NSTG - 1; ++u) issue(u);
  LTRACE(1);

  // Two copies of the K loop, selected once per workgroup: with and without the T side product.  (A uniform branch around
  // the T MFMAs INSIDE the k-step measured 19.5 -> 28.9 us on 32768x320x320: it cuts the compiler's ds_read / MFMA
  // software pipeline in two.)
  auto mainloop = [&](auto lora_tag) {
    constexpr bool LORA = decltype(lora_tag)::value;
    int rd = 0, wr = NSTG - 1;
    for (int kt = 0; kt < kt_end; ++kt) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTG - 2) * NLD) : "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
#ifdef AQL_TRACE_L
      if (kt == 0) LTRACE(2);
#endif
      issue(wr);
      const char* sA = lds + rd * STAGE;
      const char* sB = sA + A_BYTES;
      const char* sL = sB + B_BYTES;
#pragma unroll
      for (int ks = 0; ks < BK / 32; ++ks) {
        bf16x8_t fa[FM], fb[FN], fl[FT];
        const int chunk = ks * 4 + (lane >> 4);
#pragma unroll
        for (int i = 0; i < FM; ++i)
          fa[i] = *reinterpret_cast<const bf16x8_t*>(sA + lds_off(wm0 + i * 16 + (lane & 15), chunk));
#pragma unroll
        for (int j = 0; j < FN; ++j)
          fb[j] = *reinterpret_cast<const bf16x8_t*>(sB + lds_off(wn0 + j * 16 + (lane & 15), chunk));
        if constexpr (LORA) {
#pragma unroll
          for (int t = 0; t < FT; ++t)
            fl[t] = *reinterpret_cast<const bf16x8_t*>(sL + lds_off((wt0 + t) * 16 + (lane & 15), chunk));
        }
#pragma unroll
        for (int i = 0; i < FM; ++i) {
#pragma unroll
          for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
          if constexpr (LORA) {
#pragma unroll
            for (int t = 0; t < FT; ++t) tacc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fl[t], fa[i], tacc[i][t], 0, 0, 0);
          }
        }
      }
      rd = (rd + 1 == NSTG) ? 0 : rd + 1;
      wr = (wr + 1 == NSTG) ? 0 : wr + 1;
    }
  };
  if (lora_on) mainloop(std::true_type{});
  else mainloop(std::false_type{});
  LTRACE(3);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the trailing zero-fill DMAs still write LDS
  __syncthreads();
  if (lora_on) {

  // ---- T -> (T, Ts) bf16; Ts into the stage-0 A tile (fragment layout), the Bup panel into the stage-0 B tile
  // tacc[i][t][e]: row m = m0 + wm0 + 16 i + (lane & 15), rank column r = 16 (wt0 + t) + 4 (lane >> 4) + e
  char* sA = lds;
  char* sB = lds + A_BYTES;
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int row = wm0 + i * 16 + (lane & 15);
    const long m = (long)m0 + row;
    const bool ok = m < g.M;
#pragma unroll
    for (int t = 0; t < FT; ++t) {
      const int r = (wt0 + t) * 16 + (lane >> 4) * 4;
      const uint2 tv = make_uint2(pack_bf16x2(tacc[i][t][0], tacc[i][t][1]), pack_bf16x2(tacc[i][t][2], tacc[i][t][3]));
      const uint2 sv = srow[i][t];
      const uint2 ts = make_uint2(pack_bf16x2(bf16lo(tv.x) * bf16lo(sv.x), bf16hi(tv.x) * bf16hi(sv.x)),
                                  pack_bf16x2(bf16lo(tv.y) * bf16lo(sv.y), bf16hi(tv.y) * bf16hi(sv.y)));
      *reinterpret_cast<uint2*>(sA + lds_off(row, r >> 3) + (r & 7) * 2) = ts;
      if (ok && t_writer) {
        *reinterpret_cast<uint2*>(Tg + m * LR + r) = tv;
        *reinterpret_cast<uint2*>(Tsg + m * LR + r) = ts;
      }
    }
  }
#pragma unroll
  for (int u = 0; u < NBP; ++u) {
    const int id = tid + u * NTHREADS, row = id >> 2, c = id & 3;
    if (id < BN * 4) *reinterpret_cast<uint4*>(sB + lds_off(row, c)) = bup[u];
  }
  __syncthreads();
  {
    bf16x8_t fa[FM], fb[FN];
    const int chunk = lane >> 4;  // the 32 rank columns are k-step 0 of the tile
#pragma unroll
    for (int i = 0; i < FM; ++i)
      fa[i] = *reinterpret_cast<const bf16x8_t*>(sA + lds_off(wm0 + i * 16 + (lane & 15), chunk));
#pragma unroll
    for (int j = 0; j < FN; ++j)
      fb[j] = *reinterpret_cast<const bf16x8_t*>(sB + lds_off(wn0 + j * 16 + (lane & 15), chunk));
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
  }
  __syncthreads();
  }  // lora_on
  LTRACE(4);

  // ---- epilogue (as gemm_body_d, EPI_BF16): bias in fp32, C tile staged through LDS, residual added in bf16
  const EpiParams& ep = g.epi;
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int row = wm0 + i * 16 + (lane & 15);
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int col = wn0 + j * 16 + (lane >> 4) * 4;
      float v0 = acc[i][j][0], v1 = acc[i][j][1], v2 = acc[i][j][2], v3 = acc[i][j][3];
      v0 += bf16lo(biasr[j].x);
      v1 += bf16hi(biasr[j].x);
      v2 += bf16lo(biasr[j].y);
      v3 += bf16hi(biasr[j].y);
      *reinterpret_cast<uint2*>(lds + row * C_PITCH + col * 2) = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
    }
  }
  __syncthreads();
  LTRACE(5);
  if (gF) geglu_store<BM, BN, C_PITCH, NTHREADS>(lds, m0, n0, g.M, ep, tid);
  else epi_store_tile<BM, BN, C_PITCH, NTHREADS>(lds, m0, n0, g.M, g.N, ep, tid);
#ifdef AQL_TRACE_L
  LTRACE(6);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  LTRACE(7);
#endif
}

// (A one-stage, three-workgroups-per-CU form of the kernel above -- 43 KB of LDS, 168 VGPRs, the tile waited for in every k-step --
// was built and measured in round 4: bit-identical, 1.05-1.2x SLOWER on 16 of 18 shapes, profiles/r04_lora_three_workgroups.txt; its
// source is at commit 1d72eae.  A third dependent chain per CU does not make up for losing the workgroup's own DMA / MFMA overlap.)
#ifdef AQL_EXPERIMENTS   // shelved in round 4 (measured equal or slower); tools/build_alt.sh -DAQL_EXPERIMENTS builds it for A/B runs
// EXPERIMENT, off by default (AQL_LORA_PERSIST / AQL_LORA_CFG=p128; measured equal or slower, see the launcher) --
// persistent form of the 4-wave kernel for grids of several chip-wide rounds (ff.net.0 + GEGLU at the 64x64 level: 4096 tiles
// of 128x160 = 8 rounds of two workgroups per CU).  A workgroup of the one-shot kernel spends 5-7k of its ~30k cycles before its
// first MFMA (tile arithmetic, stager set-up, the first-touch latency of its first K tile: tools/trace_lora.py) and the same
// again per round.  Here a workgroup walks the tiles  first, first + G, first + 2G ...  (G = grid size, round-major, each round
// XCD-remapped like the one-shot grid), and in the LAST k-step of a tile -- where the one-shot kernel issues a zero-fill DMA
// into the free ring stage -- it re-positions its stagers on the NEXT tile and issues that tile's first K tile instead, so
// that it lands under the LoRA up step and the epilogue.  Two-stage ring (80 KB: two workgroups per CU); the bf16 C tile is
// staged in two 64-row halves through the stage the last k-step used, the other stage holds the next tile's data.  Barriers in
// the tail are raw `s_barrier`s behind `lgkmcnt(0)` only: `__syncthreads()` would drain the LDS-DMA in flight (`vmcnt(0)`).
// Same arithmetic in the same order as lora_gemm_kernel: bit-identical outputs (tools/probe_lora_gemm.py, probe_geglu.py).
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(NTHREADS, 2) void lora_gemm_kernel_p(const GemmArgs<PlainLoader, PlainLoader> g, const PlainLoader la,
                                                               const LoraParams lp, const int ntiles) {
  constexpr int FM = WM / 16, FN = WN / 16;
  constexpr int WAVES_N = BN / WN;
  static_assert((BM / WM) * (BN / WN) == 4 && BM / WM == 2, "2 x 2 wavefronts: each C half belongs to one wavefront row");
  constexpr int FT = LR / (16 * WAVES_N);
  static_assert(FT >= 1, "at most two wavefronts along N");
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, L_BYTES = LR * 128;
  constexpr int STAGE = A_BYTES + B_BYTES + L_BYTES;
  constexpr int C_PITCH = (BN + 8) * 2;
  constexpr int HB = BM / 2;                         // rows per C half
  static_assert(HB * C_PITCH <= STAGE, "a C half fits one ring stage");
  static_assert(2 * STAGE <= 80 * 1024, "two workgroups per CU");
  __shared__ __attribute__((aligned(16))) char lds[2 * STAGE];
#define AQL_LBAR() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)

  const int G = gridDim.x, bid = blockIdx.x;
  const int qq = G >> 3, rr = G & 7, xcd = bid & 7;
  const int first = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);   // XCD-aware remap inside a round
  if (first >= ntiles) return;
  const int tid0 = threadIdx.x;
  const int tiles_n = (g.N + BN - 1) / BN, tiles_m = (g.M + BM - 1) / BM;
  const int gF = g.epi.geglu_F;
  const int kt_end = g.ktiles0;
  const EpiParams& ep = g.epi;

  struct Tile { int m0, n0, grp; bool lora_on, t_writer; };
  auto tile_of = [&](int t) {
    Tile c;
    int tile_m, tile_n;
    if (g.m_fast) {
      tile_n = t / tiles_m;
      tile_m = t - tile_n * tiles_m;
    } else {
      tile_m = t / tiles_n;
      tile_n = t - tile_m * tiles_n;
    }
    c.m0 = tile_m * BM;
    c.n0 = tile_n * (gF ? BN / 2 : BN);
    c.grp = 0;
    if (lp.ngroups > 0)
      while (c.grp + 1 < lp.ngroups && c.n0 >= lp.col_start[c.grp + 1]) ++c.grp;
    c.t_writer = lp.ngroups > 0 ? (c.n0 == lp.col_start[c.grp]) : (tile_n == 0);
    c.lora_on = c.m0 + BM > lp.row0;
    return c;
  };

  DmaStager<BM, PlainLoader> sa;
  DmaStager<BN, PlainLoader> sb;
  DmaStager<LR, PlainLoader> sl;
  auto position = [&](const Tile& c, int tid) {   // stagers on K tile 0 of tile c
    PlainLoader lag = la;
    lag.base += (long)c.grp * LR * la.ld;
    if (!c.lora_on) lag.rows = 0;        // every row out of range: the A DMAs are zero fills without traffic
    sa.begin(g.a0, g.a0, false, c.m0, tid, 0, kt_end, kt_end);
    sb.begin(g.b0, g.b0, false, c.n0, tid, 0, kt_end, kt_end);
    sl.begin(lag, lag, false, 0, tid, 0, kt_end, kt_end);
  };
  auto issue = [&](int stage, int wave) {
    char* sA = lds + stage * STAGE;
    sa.dma(sA, wave);
    sb.dma(sA + A_BYTES, wave);
    sl.dma(sA + A_BYTES + B_BYTES, wave);
  };

  int t = first;
  Tile cur = tile_of(t);
  position(cur, tid0);
  issue(0, tid0 >> 6);
  int rd = 0;   // ring stage that holds K tile 0 of the current tile

  for (;;) {
    // lane-derived values are re-derived per tile from a laundered thread id: hoisted out of the tile loop (LICM) the fragment /
    // epilogue address registers of ALL phases stay live across the whole body -- 41 spilled VGPRs at two workgroups per CU
    int tid = tid0;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * WN;
    const int wt0 = (wave % WAVES_N) * FT;
    const int tn = t + G;
    const bool has_next = tn < ntiles;
    const Tile nxt = tile_of(has_next ? tn : t);
    const int m0 = cur.m0, n0 = cur.n0;
    const bool lora_on = cur.lora_on;
    bf16_t* const Tg = lp.T + (long)cur.grp * g.M * LR;
    bf16_t* const Tsg = lp.Ts + (long)cur.grp * g.M * LR;

    f32x4_t acc[FM][FN], tacc[FM][FT];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int tt = 0; tt < FT; ++tt) tacc[i][tt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    // this tile's bias values, up-projection panel and scale rows are needed after the K loop: they are requested in the LAST
    // k-step, in front of the next tile's DMA (loads retire in order: behind it they would wait for that HBM-cold tile), and
    // land under that k-step's MFMAs -- held across the whole K loop they cost 30 VGPRs the 2-workgroup budget does not have
    uint2 biasr[FN];
    constexpr int NBP = (BN * 4 + NTHREADS - 1) / NTHREADS;
    uint4 bup[NBP];
    uint2 srow[FM][FT];
    auto tail_loads = [&]() {
      epi_load_bias<FN>(biasr, ep.bias, g.b0.base, n0, wn0, lane, g.N, gF, BN / 2);
#pragma unroll
      for (int u = 0; u < NBP; ++u) {
        const int id = tid + u * NTHREADS, row = id >> 2, c = id & 3;
        const int brow = epi_bias_col(n0, row, gF, BN / 2);
        const bool ok = (id < BN * 4) & (brow < g.N) & lora_on;
        bup[u] = epi_mask4(*reinterpret_cast<const uint4*>(lp.Bup + (ok ? (long)brow * LR + c * 8 : 0)), ok);
      }
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const long m = (long)m0 + wm0 + i * 16 + (lane & 15);
#pragma unroll
        for (int tt = 0; tt < FT; ++tt) {
          const bool ok = (m < g.M) & lora_on;
          srow[i][tt] = epi_mask2(*reinterpret_cast<const uint2*>(lp.S + (ok ? (long)((uint32_t)m / (uint32_t)lp.rps) * LR + (wt0 + tt) * 16 + (lane >> 4) * 4 : 0)), ok);
        }
      }
    };

    auto mainloop = [&](auto lora_tag) {
      constexpr bool LORA = decltype(lora_tag)::value;
      auto kstep = [&]() {
        const char* sA = lds + rd * STAGE;
        const char* sB = sA + A_BYTES;
        const char* sL = sB + B_BYTES;
#pragma unroll
        for (int ks = 0; ks < BK / 32; ++ks) {
          bf16x8_t fa[FM], fb[FN], fl[FT];
          const int chunk = ks * 4 + (lane >> 4);
#pragma unroll
          for (int i = 0; i < FM; ++i)
            fa[i] = *reinterpret_cast<const bf16x8_t*>(sA + lds_off(wm0 + i * 16 + (lane & 15), chunk));
#pragma unroll
          for (int j = 0; j < FN; ++j)
            fb[j] = *reinterpret_cast<const bf16x8_t*>(sB + lds_off(wn0 + j * 16 + (lane & 15), chunk));
          if constexpr (LORA) {
#pragma unroll
            for (int tt = 0; tt < FT; ++tt)
              fl[tt] = *reinterpret_cast<const bf16x8_t*>(sL + lds_off((wt0 + tt) * 16 + (lane & 15), chunk));
          }
#pragma unroll
          for (int i = 0; i < FM; ++i) {
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
            if constexpr (LORA) {
#pragma unroll
              for (int tt = 0; tt < FT; ++tt) tacc[i][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fl[tt], fa[i], tacc[i][tt], 0, 0, 0);
            }
          }
        }
        rd ^= 1;
      };
      for (int kt = 0; kt + 1 < kt_end; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // K tile kt has landed (and the previous tile's stores are out)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue(rd ^ 1, wave);
        kstep();
      }
      // last k-step: the free stage takes the NEXT tile's first K tile
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      tail_loads();
      if (has_next) {
        position(nxt, tid);
        issue(rd ^ 1, wave);
      }
      kstep();
    };
    if (lora_on) mainloop(std::true_type{});
    else mainloop(std::false_type{});
    // rd = the stage the next tile's first K tile is landing in; rd ^ 1 = the stage of the last k-step, free from here on
    char* const scr = lds + (rd ^ 1) * STAGE;
    AQL_LBAR();   // every wavefront has read its last fragments
    if (lora_on) {
      char* sA = scr;
      char* sB = scr + A_BYTES;
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int row = wm0 + i * 16 + (lane & 15);
        const long m = (long)m0 + row;
        const bool ok = m < g.M;
#pragma unroll
        for (int tt = 0; tt < FT; ++tt) {
          const int r = (wt0 + tt) * 16 + (lane >> 4) * 4;
          const uint2 tv = make_uint2(pack_bf16x2(tacc[i][tt][0], tacc[i][tt][1]), pack_bf16x2(tacc[i][tt][2], tacc[i][tt][3]));
          const uint2 sv = srow[i][tt];
          const uint2 ts = make_uint2(pack_bf16x2(bf16lo(tv.x) * bf16lo(sv.x), bf16hi(tv.x) * bf16hi(sv.x)),
                                      pack_bf16x2(bf16lo(tv.y) * bf16lo(sv.y), bf16hi(tv.y) * bf16hi(sv.y)));
          *reinterpret_cast<uint2*>(sA + lds_off(row, r >> 3) + (r & 7) * 2) = ts;
          if (ok && cur.t_writer) {
            *reinterpret_cast<uint2*>(Tg + m * LR + r) = tv;
            *reinterpret_cast<uint2*>(Tsg + m * LR + r) = ts;
          }
        }
      }
#pragma unroll
      for (int u = 0; u < NBP; ++u) {
        const int id = tid + u * NTHREADS, row = id >> 2, c = id & 3;
        if (id < BN * 4) *reinterpret_cast<uint4*>(sB + lds_off(row, c)) = bup[u];
      }
      AQL_LBAR();
      {
        bf16x8_t fa[FM], fb[FN];
        const int chunk = lane >> 4;
#pragma unroll
        for (int i = 0; i < FM; ++i)
          fa[i] = *reinterpret_cast<const bf16x8_t*>(sA + lds_off(wm0 + i * 16 + (lane & 15), chunk));
#pragma unroll
        for (int j = 0; j < FN; ++j)
          fb[j] = *reinterpret_cast<const bf16x8_t*>(sB + lds_off(wn0 + j * 16 + (lane & 15), chunk));
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
      }
      AQL_LBAR();
    }
    // ---- epilogue in two passes of 64 rows: in pass p every wavefront stages ITS fragment rows 2p, 2p+1 (32 rows), so that half
    // of every wavefront's accumulators is dead before the first store pass needs its registers (a pass of one wavefront row
    // would keep the other row's 80 accumulator registers alive under the store loop: 33 spilled VGPRs).  LDS rows 0..31 of a
    // pass = tile rows 32p.., rows 32..63 = tile rows 64 + 32p..
    static_assert(FM == 4, "two passes of two fragment rows");
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) {
        const int i = 2 * h + ii;
        const int row = (wm0 / WM) * 32 + ii * 16 + (lane & 15);
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const int col = wn0 + j * 16 + (lane >> 4) * 4;
          float v0 = acc[i][j][0], v1 = acc[i][j][1], v2 = acc[i][j][2], v3 = acc[i][j][3];
          v0 += bf16lo(biasr[j].x);
          v1 += bf16hi(biasr[j].x);
          v2 += bf16lo(biasr[j].y);
          v3 += bf16hi(biasr[j].y);
          *reinterpret_cast<uint2*>(scr + row * C_PITCH + col * 2) = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
        }
      }
      AQL_LBAR();
#pragma unroll
      for (int q = 0; q < 2; ++q) {   // the two 32-row blocks of this pass
        const char* blk = scr + q * 32 * C_PITCH;
        const int mq = m0 + q * WM + h * 32;
        if (gF) geglu_store<32, BN, C_PITCH, NTHREADS>(blk, mq, n0, g.M, ep, tid);
        else epi_store_tile<32, BN, C_PITCH, NTHREADS>(blk, mq, n0, g.M, g.N, ep, tid);
      }
      if (h == 0) AQL_LBAR();   // the second pass overwrites the staging rows (the K loop's first barrier closes the last one)
    }
    if (!has_next) break;
    t = tn;
    cur = nxt;
  }
#undef AQL_LBAR
}
#endif   // AQL_EXPERIMENTS

// Wave-specialised form (as gemm_kernel_w in aql_gemm.cuh): 512 threads, wavefronts 4-7 only issue the LDS-DMA loads
// (X tile, W tile and the 32 rows of A), wavefronts 0-3 only read fragments and issue MFMAs, with the fragments of the next
// k-half fetched under the current MFMA batch.  Used when the grid is about one workgroup per CU and K >= 8 tiles; on the
// plain GEMM that form measured 16-18 % faster than the 4-wave kernel on exactly the shapes the LoRA linears have
// (1024x1280x1280: 12.7 vs 15.4 us, 4096x640x640: 12.9 vs 15.3 us).
template <int BM, int BN, int WM, int WN, int NSTG>
__global__ __launch_bounds__(2 * NTHREADS) void lora_gemm_kernel_w(const GemmArgs<PlainLoader, PlainLoader> g,
                                                                   const PlainLoader la, const LoraParams lp) {
  constexpr int FM = WM / 16, FN = WN / 16;
  constexpr int WAVES_N = BN / WN;
  static_assert((BM / WM) * (BN / WN) == 4, "4 compute wavefronts (+ 4 loader wavefronts) per workgroup");
  constexpr int FT = LR / (16 * WAVES_N);
  static_assert(FT >= 1 && NSTG >= 3, "ring of >= 3 stages");
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, L_BYTES = LR * 128;
  constexpr int STAGE = A_BYTES + B_BYTES + L_BYTES;
  constexpr int C_PITCH = (BN + 8) * 2;
  constexpr int LDS_BYTES = (NSTG * STAGE > BM * C_PITCH) ? NSTG * STAGE : BM * C_PITCH;
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
  __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];

  const int nblk = gridDim.x, bid = blockIdx.x;
  const int qq = nblk >> 3, rr = nblk & 7, xcd = bid & 7;
  const int block_x = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wave >= 4;
  const int ltid = tid & 255;
  const int wm0 = ((wave & 3) / WAVES_N) * WM, wn0 = ((wave & 3) % WAVES_N) * WN;
  const int wt0 = ((wave & 3) % WAVES_N) * FT;
  const int tiles_n = (g.N + BN - 1) / BN, tiles_m = (g.M + BM - 1) / BM;
  int tile_m, tile_n;
  if (g.m_fast) {
    tile_n = block_x / tiles_m;
    tile_m = block_x - tile_n * tiles_m;
  } else {
    tile_m = block_x / tiles_n;
    tile_n = block_x - tile_m * tiles_n;
  }
  const int gF = g.epi.geglu_F;  // GEGLU tiles: [80 value | 80 gate] columns (aql_gemm.cuh)
  const int m0 = tile_m * BM, n0 = tile_n * (gF ? BN / 2 : BN);
  const int kt_end = g.ktiles0;
  constexpr int NLD = BM / 32 + BN / 32 + 1;
  int grp = 0;
  if (lp.ngroups > 0)
    while (grp + 1 < lp.ngroups && n0 >= lp.col_start[grp + 1]) ++grp;
  const bool t_writer = lp.ngroups > 0 ? (n0 == lp.col_start[grp]) : (tile_n == 0);
  bf16_t* const Tg = lp.T + (long)grp * g.M * LR;
  bf16_t* const Tsg = lp.Ts + (long)grp * g.M * LR;
  PlainLoader lag = la;
  lag.base += (long)grp * LR * la.ld;
  const bool lora_on = m0 + BM > lp.row0;   // block-uniform
  if (!lora_on) lag.rows = 0;

  f32x4_t acc[FM][FN], tacc[FM][FT];
#pragma unroll
  for (int i = 0; i < FM; ++i) {
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < FT; ++t) tacc[i][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }
  constexpr int NBP = (BN * 4 + NTHREADS - 1) / NTHREADS;
  uint4 bup[NBP];          // loader wavefronts: the Bup panel, fetched before the K loop
  uint2 srow[FM][FT];      // compute wavefronts: this lane's scale rows
  uint2 biasr[FN];         // compute wavefronts: this lane's bias values (epi_load_bias, aql_gemm.cuh)

  if (loader) {
#pragma unroll
    for (int u = 0; u < NBP; ++u) {
      const int id = ltid + u * NTHREADS, row = id >> 2, c = id & 3;
      const int brow = epi_bias_col(n0, row, gF, BN / 2);
      const bool ok = (id < BN * 4) & (brow < g.N) & lora_on;   // unconditional load, clamped address, AND-mask
      bup[u] = epi_mask4(*reinterpret_cast<const uint4*>(lp.Bup + (ok ? (long)brow * LR + c * 8 : 0)), ok);
    }
    DmaStager<BM, PlainLoader> sa;
    DmaStager<BN, PlainLoader> sb;
    DmaStager<LR, PlainLoader> sl;
    sa.begin(g.a0, g.a0, false, m0, ltid, 0, kt_end, kt_end);
    sb.begin(g.b0, g.b0, false, n0, ltid, 0, kt_end, kt_end);
    sl.begin(lag, lag, false, 0, ltid, 0, kt_end, kt_end);
    auto issue = [&](int stage) {
      char* sA = lds + stage * STAGE;
      sa.dma(sA, wave - 4);
      sb.dma(sA + A_BYTES, wave - 4);
      sl.dma(sA + A_BYTES + B_BYTES, wave - 4);
    };
#pragma unroll
    for (int u = 0; u < NSTG - 1; ++u) issue(u);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTG - 2) * NLD) : "memory");  // first tile landed (the Bup loads are older)
    __builtin_amdgcn_s_barrier();
    int wr = NSTG - 1;
    for (int kt = 0; kt < kt_end; ++kt) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTG - 3) * NLD) : "memory");  // tile kt+1 landed
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      issue(wr);
      wr = (wr + 1 == NSTG) ? 0 : wr + 1;
    }
  } else {
    epi_load_bias<FN>(biasr, g.epi.bias, g.b0.base, n0, wn0, lane, g.N, gF, BN / 2);
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const long m = (long)m0 + wm0 + i * 16 + (lane & 15);
#pragma unroll
      for (int t = 0; t < FT; ++t) {
        const bool ok = (m < g.M) & lora_on;
        srow[i][t] = epi_mask2(*reinterpret_cast<const uint2*>(lp.S + (ok ? (long)((uint32_t)m / (uint32_t)lp.rps) * LR + (wt0 + t) * 16 + (lane >> 4) * 4 : 0)), ok);
      }
    }
    const int arow = wm0 + (lane & 15), brow = wn0 + (lane & 15), lrow = wt0 * 16 + (lane & 15);
    const int ch0 = lane >> 4, ch1 = 4 + (lane >> 4);
    bf16x8_t fa0[FM], fb0[FN], fl0[FT], fa1[FM], fb1[FN], fl1[FT];
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int j = 0; j < FN; ++j) fb0[j] = *reinterpret_cast<const bf16x8_t*>(lds + A_BYTES + lds_off(brow + j * 16, ch0));
#pragma unroll
    for (int t = 0; t < FT; ++t) fl0[t] = *reinterpret_cast<const bf16x8_t*>(lds + A_BYTES + B_BYTES + lds_off(lrow + t * 16, ch0));
#pragma unroll
    for (int i = 0; i < FM; ++i) fa0[i] = *reinterpret_cast<const bf16x8_t*>(lds + lds_off(arow + i * 16, ch0));
    int rd = 0;
    for (int kt = 0; kt < kt_end; ++kt) {
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const char* sA = lds + rd * STAGE;
      const char* sB = sA + A_BYTES;
      const char* sL = sB + B_BYTES;
      rd = (rd + 1 == NSTG) ? 0 : rd + 1;
      const char* nA = lds + rd * STAGE;
      const char* nB = nA + A_BYTES;
      const char* nL = nB + B_BYTES;
#pragma unroll
      for (int j = 0; j < FN; ++j) fb1[j] = *reinterpret_cast<const bf16x8_t*>(sB + lds_off(brow + j * 16, ch1));
#pragma unroll
      for (int t = 0; t < FT; ++t) fl1[t] = *reinterpret_cast<const bf16x8_t*>(sL + lds_off(lrow + t * 16, ch1));
#pragma unroll
      for (int i = 0; i < FM; ++i) fa1[i] = *reinterpret_cast<const bf16x8_t*>(sA + lds_off(arow + i * 16, ch1));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < FM; ++i) {
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb0[j], fa0[i], acc[i][j], 0, 0, 0);
      }
      if (lora_on) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int t = 0; t < FT; ++t) tacc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fl0[t], fa0[i], tacc[i][t], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < FN; ++j) fb0[j] = *reinterpret_cast<const bf16x8_t*>(nB + lds_off(brow + j * 16, ch0));
#pragma unroll
      for (int t = 0; t < FT; ++t) fl0[t] = *reinterpret_cast<const bf16x8_t*>(nL + lds_off(lrow + t * 16, ch0));
#pragma unroll
      for (int i = 0; i < FM; ++i) fa0[i] = *reinterpret_cast<const bf16x8_t*>(nA + lds_off(arow + i * 16, ch0));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < FM; ++i) {
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb1[j], fa1[i], acc[i][j], 0, 0, 0);
      }
      if (lora_on) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int t = 0; t < FT; ++t) tacc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fl1[t], fa1[i], tacc[i][t], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  char* sA = lds;
  char* sB = lds + A_BYTES;
  if (lora_on) {
  if (loader) {
#pragma unroll
    for (int u = 0; u < NBP; ++u) {
      const int id = ltid + u * NTHREADS, row = id >> 2, c = id & 3;
      if (id < BN * 4) *reinterpret_cast<uint4*>(sB + lds_off(row, c)) = bup[u];
    }
  } else {
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int row = wm0 + i * 16 + (lane & 15);
      const long m = (long)m0 + row;
      const bool ok = m < g.M;
#pragma unroll
      for (int t = 0; t < FT; ++t) {
        const int r = (wt0 + t) * 16 + (lane >> 4) * 4;
        const uint2 tv = make_uint2(pack_bf16x2(tacc[i][t][0], tacc[i][t][1]), pack_bf16x2(tacc[i][t][2], tacc[i][t][3]));
        const uint2 sv = srow[i][t];
        const uint2 ts = make_uint2(pack_bf16x2(bf16lo(tv.x) * bf16lo(sv.x), bf16hi(tv.x) * bf16hi(sv.x)),
                                    pack_bf16x2(bf16lo(tv.y) * bf16lo(sv.y), bf16hi(tv.y) * bf16hi(sv.y)));
        *reinterpret_cast<uint2*>(sA + lds_off(row, r >> 3) + (r & 7) * 2) = ts;
        if (ok && t_writer) {
          *reinterpret_cast<uint2*>(Tg + m * LR + r) = tv;
          *reinterpret_cast<uint2*>(Tsg + m * LR + r) = ts;
        }
      }
    }
  }
  __syncthreads();
  if (!loader) {
    bf16x8_t fa[FM], fb[FN];
    const int chunk = lane >> 4;
#pragma unroll
    for (int i = 0; i < FM; ++i)
      fa[i] = *reinterpret_cast<const bf16x8_t*>(sA + lds_off(wm0 + i * 16 + (lane & 15), chunk));
#pragma unroll
    for (int j = 0; j < FN; ++j)
      fb[j] = *reinterpret_cast<const bf16x8_t*>(sB + lds_off(wn0 + j * 16 + (lane & 15), chunk));
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
  }
  __syncthreads();
  }  // lora_on

  const EpiParams& ep = g.epi;
  if (!loader) {
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int row = wm0 + i * 16 + (lane & 15);
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int col = wn0 + j * 16 + (lane >> 4) * 4;
        float v0 = acc[i][j][0], v1 = acc[i][j][1], v2 = acc[i][j][2], v3 = acc[i][j][3];
        v0 += bf16lo(biasr[j].x);
        v1 += bf16hi(biasr[j].x);
        v2 += bf16lo(biasr[j].y);
        v3 += bf16hi(biasr[j].y);
        *reinterpret_cast<uint2*>(lds + row * C_PITCH + col * 2) = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
      }
    }
  }
  __syncthreads();
  if (gF) geglu_store<BM, BN, C_PITCH, 2 * NTHREADS>(lds, m0, n0, g.M, ep, tid);
  else epi_store_tile<BM, BN, C_PITCH, 2 * NTHREADS>(lds, m0, n0, g.M, g.N, ep, tid);
}

template <int BM, int BN, int WM, int WN, int NSTG>
void launch_w(const GemmArgs<PlainLoader, PlainLoader>& g, const PlainLoader& la, const LoraParams& lp, hipStream_t stream) {
  dim3 grid(aql_cdiv(g.M, BM) * aql_cdiv(g.N, BN));
  hipLaunchKernelGGL((lora_gemm_kernel_w<BM, BN, WM, WN, NSTG>), grid, dim3(2 * NTHREADS), 0, stream, g, la, lp);
}

#ifdef AQL_EXPERIMENTS
// persistent 4-wave kernel: `wgs` resident workgroups (two per CU) walk all tiles
template <int BM, int BN, int WM, int WN>
void launch_p(const GemmArgs<PlainLoader, PlainLoader>& g, const PlainLoader& la, const LoraParams& lp, int wgs, hipStream_t stream) {
  const int ntiles = aql_cdiv(g.M, BM) * aql_cdiv(g.N, BN);
  dim3 grid(ntiles < wgs ? ntiles : wgs);
  hipLaunchKernelGGL((lora_gemm_kernel_p<BM, BN, WM, WN>), grid, dim3(NTHREADS), 0, stream, g, la, lp, ntiles);
}
#endif

template <int BM, int BN, int WM, int WN, int NSTG>
void launch(const GemmArgs<PlainLoader, PlainLoader>& g, const PlainLoader& la, const LoraParams& lp, hipStream_t stream) {
  dim3 grid(aql_cdiv(g.M, BM) * aql_cdiv(g.N, BN));
  hipLaunchKernelGGL((lora_gemm_kernel<BM, BN, WM, WN, NSTG>), grid, dim3(NTHREADS), 0, stream, g, la, lp);
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

// The picker of the 256 x 256 GEGLU tile, shared by the rank-32 one-launch form below and by aql_gemm_bf16_geglu (aql_gemm.hip: any
// rank as a second K segment, or no LoRA at all): taken from two chip-wide rounds of tiles on (measured: 0.71-0.90x of the 128 x 160
// kernel's time from 640 tiles up, 1.14-1.46x below 320: profiles/r04_t256_geglu.txt).  AQL_LORA_T256=n moves the threshold
// (0 = never), AQL_LORA_CFG=t256 forces the tile on every shape it can run, any other AQL_LORA_CFG keeps it off.
static bool t256_wanted(long M, int F, long ld_max, long ldw_max, long ldg) {
  static const int min_tiles = AQL_TUNE_INT("AQL_LORA_T256", 512);
  const char* cfg = getenv("AQL_LORA_CFG");
  const bool forced = cfg && cfg[0] == 't';
  const bool fits = F > 0 && F % 128 == 0 && ldg % 8 == 0 && M * ld_max < (1L << 31) && 2L * F * ldw_max < (1L << 31);   // 32-bit buffer ranges
  const long tiles = (long)aql_cdiv((int)M, 256) * (F / 128);
  return fits && !(cfg && !forced) && (forced || (min_tiles > 0 && tiles >= min_tiles));
}

int aqlt256::t256_geglu_two_segments(const bf16_t* A, long lda, const bf16_t* B, long ldb, long M, int F, int K, const bf16_t* A2,
                                     long lda2, const bf16_t* B2, long ldb2, int K2, const bf16_t* bias, bf16_t* H, long ldh, bf16_t* G,
                                     long ldg, long row0, hipStream_t stream) {
  const long ldm = std::max(std::max(lda, A2 ? lda2 : 0L), H ? ldh : 0L);
  if (!t256_wanted(M, F, ldm, std::max(ldb, B2 ? ldb2 : 0L), ldg)) return AQL_NOT_FUSED;
  aqlt256::Args a{};
  a.X = A, a.W = B, a.bias = bias, a.H = H, a.G = G;
  a.ldx = lda, a.ldw = ldb, a.ldh = ldh, a.ldg = ldg;
  a.M = (int)M, a.K = K, a.F = F, a.rps = (int)M;
  a.row0 = (int)(row0 < 0 ? 0 : (row0 > M ? M : row0));
  a.c_row0 = a.row0;
  a.ntiles = aql_cdiv((int)M, 256) * (F / 128);
  a.seg2 = 1;
  if (A2 != nullptr) a.X2 = A2, a.W2 = B2, a.ldx2 = lda2, a.ldw2 = ldb2, a.K2 = K2;
  aqlt256::launch(a, stream);
  AQL_CHECK_LAUNCH("aql_gemm_bf16_geglu (256 x 256 tile)");
  return AQL_OK;
}

// Y[M,N] = X[M,K].W[N,K]^T + ((X.A[32,K]^T) * S[m / rps]).Bup[N,32]^T + bias + residual;  T, Ts [M,32] are written too.
// Returns AQL_OK, an error, or AQL_NOT_FUSED (100) when the shape belongs on the two-launch path (deep K on a small grid:
// split-K; N <= 32).
static int lora_gemm_fused_impl(const bf16_t* X, long ldx, const bf16_t* W, long ldw, long M, int N, int K,
                                const bf16_t* Adown, const bf16_t* S, int rows_per_sample, const bf16_t* Bup,
                                const bf16_t* bias, const bf16_t* residual, long ldr, bf16_t* Y, long ldy, bf16_t* T,
                                bf16_t* Ts, bf16_t* G, long ldg, int geglu_F, int ngroups, const int* col_start, long row0,
                                hipStream_t stream, const bf16_t* gb_h = nullptr, long gb_ldh = 0) {
  AQL_CHECK_ARG(X && W && Adown && S && Bup && (Y || geglu_F) && T && Ts, "aql_lora_gemm_fused: null operand");
  AQL_CHECK_ARG(M * ldx * 2 < (long)BUF_BYTES && (long)N * ldw * 2 < (long)BUF_BYTES,
                "aql_lora_gemm_fused: an operand spans >= 1 GiB, the buffer descriptors cover %u bytes (split the rows)", BUF_BYTES);
  AQL_CHECK_ARG(M > 0 && M < (1L << 31) && N > 0 && K > 0 && N % 8 == 0 && K % 8 == 0 && ldx % 8 == 0 && ldw % 8 == 0 &&
                    ldy % 8 == 0 && rows_per_sample > 0 && (residual == nullptr || ldr % 8 == 0),
                "aql_lora_gemm_fused: bad shape M=%ld N=%d K=%d", M, N, K);
  if (N <= 32) return AQL_NOT_FUSED;
  const int kt = aql_cdiv(K, BK);
  // Short K under a very wide N (ff.net.0 at the 64x64 / 32x32 levels: 320 -> 2560, 640 -> 5120): 16-32 column tiles each
  // recompute T for a 5-10 tile K loop, and in isolation the two-launch form is faster there (58.3 vs 66.9 us, 50.0 vs
  // 51.2 us, tools/tune_lora_gemm.py) -- but inside the step the saved launch still wins (27.65 vs 27.9 ms/step measured
  // on one box), so they are fused too; AQL_LORA_WIDE=0 restores the exclusion.
  static const int wide_ok = AQL_TUNE_INT("AQL_LORA_WIDE", 1);  // tuning hook
  if (!wide_ok && kt <= 10 && N >= 2560) return AQL_NOT_FUSED;
  GemmArgs<PlainLoader, PlainLoader> g;
  g.a0 = plain(X, ldx, M, K);
  g.b0 = plain(W, ldw, N, K);
  if (geglu_F) g.b0.gsplit = 80, g.b0.goff = geglu_F - 80;
  g.a1 = g.a0;
  g.b1 = g.b0;
  g.ktiles0 = kt;
  g.ktiles1 = 0;
  g.M = (int)M;
  g.N = N;
  g.splits = 1;
  g.m_fast = ((long)N * kt > (long)M * (kt < 9 ? kt : kt / 9 + 1)) ? 1 : 0;
  g.epi = EpiParams{};
  g.epi.C = Y;
  g.epi.ldc = ldy;
  g.epi.bias = bias;
  g.epi.residual = residual;
  g.epi.ldr = ldr;
  g.epi.rows_per_sample = rows_per_sample;
  g.epi.G = G;
  g.epi.ldg = ldg;
  g.epi.geglu_F = geglu_F;
  if (gb_h != nullptr) g.epi.gb_h = gb_h, g.epi.gb_ldh = gb_ldh, g.epi.gb_F = N;
#ifdef AQL_TRACE_L
  if (const char* tb = getenv("AQL_TRACE_BUF")) g.epi.Cf = reinterpret_cast<float*>(strtoull(tb, nullptr, 0));
#endif
  const PlainLoader la = plain(Adown, K, LR, K);
  LoraParams lp{};
  lp.S = S, lp.Bup = Bup, lp.T = T, lp.Ts = Ts, lp.rps = rows_per_sample;
  lp.row0 = (int)(row0 < 0 ? 0 : (row0 > M ? M : row0));
  g.epi.c_row0 = lp.row0;
  lp.ngroups = ngroups;
  for (int i = 0; i <= ngroups && ngroups > 0; ++i) lp.col_start[i] = col_start[i];
  if (ngroups > 0 && N % 160 != 0) return AQL_NOT_FUSED;  // groups are cut at 160-column tile boundaries
  // 256 x 256 persistent tile for the GEGLU form (aql_gemm_lora_t256.cuh)
  {
    const bool fits = geglu_F > 0 && gb_h == nullptr && ngroups == 0 && residual == nullptr && G != nullptr &&
                      rows_per_sample >= 32;   // the tile's scale-row table holds 16 samples
    if (fits && t256_wanted(M, geglu_F, ldy > ldx ? ldy : ldx, ldw, ldg)) {
      aqlt256::Args a{};
      a.X = X, a.W = W, a.Ad = Adown, a.S = S, a.Bup = Bup, a.bias = bias;
      a.H = Y, a.G = G, a.T = T, a.Ts = Ts;
      a.ldx = ldx, a.ldw = ldw, a.ldh = ldy, a.ldg = ldg;
      a.M = (int)M, a.K = K, a.F = geglu_F, a.rps = rows_per_sample, a.row0 = lp.row0, a.c_row0 = lp.row0;
      a.ntiles = aql_cdiv((int)M, 256) * (geglu_F / 128);
#ifdef AQL_T256_TRACE
      if (const char* tb = getenv("AQL_TRACE_BUF")) a.trace = reinterpret_cast<long long*>(strtoull(tb, nullptr, 0));
#endif
      aqlt256::launch(a, stream);
      AQL_CHECK_LAUNCH("aql_lora_gemm_fused (256 x 256 tile)");
      return AQL_OK;
    }
  }
  static const int deep_kt = AQL_TUNE_INT("AQL_DEEPKT", 32);
  static const int force_bm = AQL_TUNE_INT("AQL_LORA_BM", 0);  // tuning hook (160-wide tiles)
  // tuning hook (tools/tune_lora_cfg.py), re-read on every call: w128 / w64 / w32 = wave-specialised kernel with that tile
  // height, d128 / d64 / d32 = 4-wave kernel (ring depth by grid size), d128s / d64s / d32s = 4-wave kernel, 2 stages
  if (const char* cfg = getenv("AQL_LORA_CFG")) {
    if (N % 160 == 0 && cfg[0]) {
      const int bm = atoi(cfg + 1);
      const int tiles = aql_cdiv(M, bm) * (N / 160);
      const bool shallow = cfg[strlen(cfg) - 1] == 's' || tiles > 288;
      bool ok = true;
#ifdef AQL_EXPERIMENTS
      if (cfg[0] == 'p' && bm == 128) launch_p<128, 160, 64, 80>(g, la, lp, 512, stream);
      else
#endif
      if (cfg[0] == 'w' && bm == 128) launch_w<128, 160, 64, 80, 3>(g, la, lp, stream);
      else if (cfg[0] == 'w' && bm == 64) launch_w<64, 160, 32, 80, 4>(g, la, lp, stream);
      else if (cfg[0] == 'w' && bm == 32) launch_w<32, 160, 16, 80, 5>(g, la, lp, stream);
      else if (cfg[0] == 'd' && bm == 128) { if (shallow) launch<128, 160, 64, 80, 2>(g, la, lp, stream); else launch<128, 160, 64, 80, 3>(g, la, lp, stream); }
      else if (cfg[0] == 'd' && bm == 64) { if (shallow) launch<64, 160, 32, 80, 2>(g, la, lp, stream); else launch<64, 160, 32, 80, 5>(g, la, lp, stream); }
      else if (cfg[0] == 'd' && bm == 32) { if (shallow) launch<32, 160, 16, 80, 2>(g, la, lp, stream); else launch<32, 160, 16, 80, 5>(g, la, lp, stream); }
      else ok = false;
      if (ok) {
        AQL_CHECK_LAUNCH("aql_lora_gemm_fused");
        return AQL_OK;
      }
    }
  }
  // tile choice: the largest 160-wide (or square) tile that still gives about one workgroup per CU
  if (N % 160 == 0) {
    const int nt = N / 160;
    const int t128 = aql_cdiv(M, 128) * nt, t64 = aql_cdiv(M, 64) * nt, t32 = aql_cdiv(M, 32) * nt;
    const int tiles = t128 >= 448 ? t128 : t64 >= 200 ? t64 : t32;
    if (tiles < 224 && kt >= deep_kt) return AQL_NOT_FUSED;  // the two-launch path would split K here
    // Deep K under few rows (d(ff.net.0) at the 16x16 level: 1024 x 1280 x 10240): the only one-launch grid that fills the chip
    // is 32-row tiles, whose weight panel traffic (3.3 MB per workgroup) makes the launch L2-bound -- 86 us against 47 us of the
    // plain split-K GEMM on 128-row tiles (tools/cmp_lora_paths.py).  With aql_lora_down_splitk the two-launch form costs
    // GEMM + ~8 us there.  AQL_LORA_DEEP_T128 = 0 restores the one-launch choice.
    static const int deep_t128 = AQL_TUNE_INT("AQL_LORA_DEEP_T128", 100);
    if (kt >= deep_kt && t128 < deep_t128 && !geglu_F && ngroups == 0 && gb_h == nullptr) return AQL_NOT_FUSED;
    static const int use_w = AQL_TUNE_INT("AQL_LORA_W", 1);
    // wave-specialised kernels: ONE chip-wide round of 8-wave workgroups (two rounds of the 128-row tile measured slower than
    // the 4-wave kernel: 52.2 vs 40.9 us at 1024x10240x1280)
    if (use_w && kt >= 8 && force_bm == 0) {
      // one round of 128-row tiles: with the weights coming from HBM (the train step: every weight is read once per pass) the
      // 4-wave kernel with a 3-stage ring is 9-15 % faster than the wave-specialised one at K >= 1280 (COLD=1 tools/tune_lora_cfg.py,
      // profiles/r02_tune_lora_cfg_cold_weights.txt); with L2-warm weights it was the other way round by 2-4 %
      static const int use_w128 = AQL_TUNE_INT("AQL_LORA_W128", 0);
      if (t128 >= 240 && t128 <= 288) {
        if (use_w128) launch_w<128, 160, 64, 80, 3>(g, la, lp, stream);
        else launch<128, 160, 64, 80, 3>(g, la, lp, stream);
        AQL_CHECK_LAUNCH("aql_lora_gemm_fused");
        return AQL_OK;
      }
      if (t128 < 240 && t64 >= 240 && t64 <= 512) { launch_w<64, 160, 32, 80, 4>(g, la, lp, stream); AQL_CHECK_LAUNCH("aql_lora_gemm_fused"); return AQL_OK; }
      if (t64 < 240 && t32 >= 240 && t32 <= 512) { launch_w<32, 160, 16, 80, 5>(g, la, lp, stream); AQL_CHECK_LAUNCH("aql_lora_gemm_fused"); return AQL_OK; }
    }
    // (round 5: 320-447 tiles of 128 rows are all resident at two workgroups per CU -- ONE round -- where the 64-row choice below
    // ran one and a half: q | k | v at the 16 x 16 level on the twin batch, 2048 x 3840 x 1280, 56.3 -> 37.2 us, tools/tune_lora_cfg.py)
    if (force_bm == 128 || (force_bm == 0 && t128 >= 320)) {
      // AQL_LORA_PERSIST=n (default 0 = off): grids of >= n tiles on the persistent kernel, whose workgroups fetch their next
      // tile's first K tile under the current tile's tail.  Measured round 4 (tools/probe_lora_persist.py, bit-identical on all 18
      // forms): 0.98x on ff.net.0 + GEGLU at 32768 x 2560 x 320 and 1.03-1.17x (SLOWER) everywhere else -- with two workgroups
      // per CU the dispatcher already starts a fresh workgroup's prologue under its neighbour's K loop (profiles/r04_lora_persistent.txt)
      if (t128 <= 288) launch<128, 160, 64, 80, 3>(g, la, lp, stream);
#ifdef AQL_EXPERIMENTS
      else if (getenv("AQL_LORA_PERSIST") && atoi(getenv("AQL_LORA_PERSIST")) > 0 && t128 >= atoi(getenv("AQL_LORA_PERSIST")))
        launch_p<128, 160, 64, 80>(g, la, lp, 512, stream);
#endif
      else launch<128, 160, 64, 80, 2>(g, la, lp, stream);
    } else if (force_bm == 64 || (force_bm == 0 && t64 >= 200)) {
      if (t64 <= 288) launch<64, 160, 32, 80, 5>(g, la, lp, stream);
      else launch<64, 160, 32, 80, 2>(g, la, lp, stream);
    } else {
      if (t32 <= 288) launch<32, 160, 16, 80, 5>(g, la, lp, stream);
      else launch<32, 160, 16, 80, 2>(g, la, lp, stream);
    }
  } else {
    if (geglu_F) return AQL_NOT_FUSED;  // the [value | gate] tile layout exists for the 160-wide tiles only
    const int t128 = aql_cdiv(M, 128) * aql_cdiv(N, 128), t64 = aql_cdiv(M, 64) * aql_cdiv(N, 64);
    const int tiles = t128 >= 240 ? t128 : t64;
    if (tiles < 224 && kt >= deep_kt) return AQL_NOT_FUSED;
    if (t128 >= 240) {
      launch<128, 128, 64, 64, 2>(g, la, lp, stream);
    } else {
      if (t64 <= 288) launch<64, 64, 32, 32, 6>(g, la, lp, stream);
      else launch<64, 64, 32, 32, 2>(g, la, lp, stream);
    }
  }
  AQL_CHECK_LAUNCH("aql_lora_gemm_fused");
  return AQL_OK;
}

extern "C" int aql_lora_gemm_fused(const bf16_t* X, long ldx, const bf16_t* W, long ldw, long M, int N, int K,
                                   const bf16_t* Adown, const bf16_t* S, int rows_per_sample, const bf16_t* Bup,
                                   const bf16_t* bias, const bf16_t* residual, long ldr, bf16_t* Y, long ldy, bf16_t* T,
                                   bf16_t* Ts, long lora_row0, hipStream_t stream) {
  return lora_gemm_fused_impl(X, ldx, W, ldw, M, N, K, Adown, S, rows_per_sample, Bup, bias, residual, ldr, Y, ldy, T, Ts,
                              nullptr, 0, 0, 0, nullptr, lora_row0, stream);
}

// Several rank-32 LoRA linears that share their input in ONE launch (q|k|v of a self-attention,
// scripts/lib/original_unet.py:688-704; the k|v projections of the text states of every cross-attention): W [N][K], Bup [N][32]
// and bias are the linears stacked along N, Adown [ngroups*32][K] the stacked down matrices, T / Ts [ngroups][M][32].
// col_start[0..ngroups] (host array): first output column of every linear, multiples of 160, col_start[ngroups] = N.
extern "C" int aql_lora_gemm_fused_grouped(const bf16_t* X, long ldx, const bf16_t* W, long ldw, long M, int N, int K,
                                           int ngroups, const int* col_start, const bf16_t* Adown, const bf16_t* S,
                                           int rows_per_sample, const bf16_t* Bup, const bf16_t* bias, bf16_t* Y, long ldy,
                                           bf16_t* T, bf16_t* Ts, long lora_row0, hipStream_t stream) {
  AQL_CHECK_ARG(ngroups >= 1 && ngroups <= MAXG && col_start != nullptr, "aql_lora_gemm_fused_grouped: 1..%d groups", MAXG);
  AQL_CHECK_ARG(col_start[0] == 0 && col_start[ngroups] == N, "aql_lora_gemm_fused_grouped: col_start must span [0, N]");
  for (int i = 0; i < ngroups; ++i)
    AQL_CHECK_ARG(col_start[i] % 160 == 0 && col_start[i] < col_start[i + 1], "aql_lora_gemm_fused_grouped: bad group %d", i);
  return lora_gemm_fused_impl(X, ldx, W, ldw, M, N, K, Adown, S, rows_per_sample, Bup, bias, nullptr, 0, Y, ldy, T, Ts, nullptr,
                              0, 0, ngroups, col_start, lora_row0, stream);
}

// ff.net.0.proj with the rank-32 watermark LoRA AND the GEGLU activation in one launch (aql_gemm_bf16_geglu's tile layout on
// the one-launch LoRA linear): W [2F][K], Bup [2F][32], bias [2F];  H [M][2F] = pre-activation (null = not written),
// G [M][F] = H[:, :F] * gelu_erf(H[:, F:]).  Returns AQL_NOT_FUSED (100) like aql_lora_gemm_fused, and when F % 80 != 0.
extern "C" int aql_lora_gemm_fused_geglu(const bf16_t* X, long ldx, const bf16_t* W, long ldw, long M, int F, int K,
                                         const bf16_t* Adown, const bf16_t* S, int rows_per_sample, const bf16_t* Bup,
                                         const bf16_t* bias, bf16_t* H, long ldh, bf16_t* G, long ldg, bf16_t* T, bf16_t* Ts,
                                         long lora_row0, hipStream_t stream) {
  AQL_CHECK_ARG(G && F > 0 && ldg % 8 == 0 && (H == nullptr || ldh % 8 == 0), "aql_lora_gemm_fused_geglu: bad args");
  if (F % 80 != 0) return AQL_NOT_FUSED;
  return lora_gemm_fused_impl(X, ldx, W, ldw, M, 2 * F, K, Adown, S, rows_per_sample, Bup, bias, nullptr, 0, H, ldh, T, Ts, G,
                              ldg, F, 0, nullptr, lora_row0, stream);
}

// Backward-data of ff.net.2 fused with the backward of the GEGLU in front of it (scripts/lib/original_unet.py:727-729):
// aql_lora_gemm_fused in its backward form (X = dY [M,K], W = W2^T [F,K], Adown = Bup2^T, Bup = A2^T: dTs, dT and
// d(activated) = dY.W2 + dT.A2 [M,F]), whose epilogue turns d(activated) into d(pre-activation) with the saved H [M,2F]:
// DH[m][n] = d * gate * cdf(gate), DH[m][F+n] = d * value * (cdf(gate) + gate pdf(gate)) -- aql_geglu_bwd applied to the
// bf16-rounded tile (bit-identical to the two kernels).  Returns 100 like aql_lora_gemm_fused.
// ng <= 3 rank-32 LoRA linears summed into ONE output (aql_gemm_lora_kgroups.cuh): the backward-data pass of q | k | v.
// X, W, Adown, Bup, T, Ts, ldx, ldw, K are HOST arrays of ng entries.  Returns 100 when no wave-specialised tile gives one
// chip-wide round for (M, N): the caller then chains aql_lora_gemm_fused launches.
extern "C" int aql_lora_gemm_fused_kgroups(int ng, const void* const* X, const long* ldx, const void* const* W, const long* ldw,
                                           const int* K, const void* const* Adown, const void* const* Bup, long M, int N,
                                           const bf16_t* S, int rows_per_sample, const bf16_t* residual, long ldr, bf16_t* Y,
                                           long ldy, void* const* T, void* const* Ts, hipStream_t stream) {
  AQL_CHECK_ARG(ng >= 1 && ng <= aqlkg::KG_MAX && X && ldx && W && ldw && K && Adown && Bup && T && Ts && S && Y,
                "aql_lora_gemm_fused_kgroups: bad operands (ng=%d)", ng);
  AQL_CHECK_ARG(M > 0 && M < (1L << 31) && N > 0 && N % 8 == 0 && ldy % 8 == 0 && rows_per_sample > 0 &&
                    (residual == nullptr || ldr % 8 == 0), "aql_lora_gemm_fused_kgroups: bad shape M=%ld N=%d", M, N);
  if (N % 160 != 0) return AQL_NOT_FUSED;
  aqlkg::KGArgs a{};
  a.ng = ng;
  int kt_sum = 0;
  for (int g = 0; g < ng; ++g) {
    AQL_CHECK_ARG(X[g] && W[g] && Adown[g] && Bup[g] && T[g] && Ts[g] && K[g] > 0 && K[g] % 8 == 0 && ldx[g] % 8 == 0 && ldw[g] % 8 == 0,
                  "aql_lora_gemm_fused_kgroups: bad group %d", g);
    a.x[g] = plain(static_cast<const bf16_t*>(X[g]), ldx[g], M, K[g]);
    a.w[g] = plain(static_cast<const bf16_t*>(W[g]), ldw[g], N, K[g]);
    a.ad[g] = plain(static_cast<const bf16_t*>(Adown[g]), K[g], LR, K[g]);
    a.bup[g] = static_cast<const bf16_t*>(Bup[g]);
    a.T[g] = static_cast<bf16_t*>(T[g]);
    a.Ts[g] = static_cast<bf16_t*>(Ts[g]);
    a.kt[g] = aql_cdiv(K[g], BK);
    kt_sum += a.kt[g];
  }
  a.S = S, a.rps = rows_per_sample, a.M = (int)M, a.N = N;
  a.m_fast = ((long)N * kt_sum > (long)M * (kt_sum < 9 ? kt_sum : kt_sum / 9 + 1)) ? 1 : 0;
  a.epi = EpiParams{};
  a.epi.C = Y, a.epi.ldc = ldy, a.epi.residual = residual, a.epi.ldr = ldr, a.epi.rows_per_sample = rows_per_sample;
  const int nt = N / 160;
  const int t128 = aql_cdiv(M, 128) * nt, t64 = aql_cdiv(M, 64) * nt, t32 = aql_cdiv(M, 32) * nt;
  if (t128 >= 240 && t128 <= 288) aqlkg::launch_wk<128, 160, 64, 80, 3>(a, stream);
  else if (t128 < 240 && t64 >= 240 && t64 <= 512) aqlkg::launch_wk<64, 160, 32, 80, 4>(a, stream);
  else if (t64 < 240 && t32 >= 240 && t32 <= 512) aqlkg::launch_wk<32, 160, 16, 80, 5>(a, stream);
  else return AQL_NOT_FUSED;
  AQL_CHECK_LAUNCH("aql_lora_gemm_fused_kgroups");
  return AQL_OK;
}

extern "C" int aql_lora_gemm_fused_geglu_bwd(const bf16_t* X, long ldx, const bf16_t* W, long ldw, long M, int F, int K,
                                             const bf16_t* Adown, const bf16_t* S, int rows_per_sample, const bf16_t* Bup,
                                             const bf16_t* H, long ldh, bf16_t* DH, long lddh, bf16_t* T, bf16_t* Ts,
                                             hipStream_t stream) {
  AQL_CHECK_ARG(H && DH && ldh % 8 == 0 && lddh % 8 == 0 && F % 8 == 0, "aql_lora_gemm_fused_geglu_bwd: bad args");
  return lora_gemm_fused_impl(X, ldx, W, ldw, M, F, K, Adown, S, rows_per_sample, Bup, nullptr, nullptr, 0, DH, lddh, T, Ts,
                              nullptr, 0, 0, 0, nullptr, 0, stream, H, ldh);
}
