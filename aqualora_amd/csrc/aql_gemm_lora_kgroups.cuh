// K-grouped form of the wave-specialised one-launch LoRA linear: ONE output  Y[M,N] = sum_g ( X_g.W_g^T + ((X_g.A_g^T) * S).Bup_g^T )
// (+ residual) over up to 3 independent rank-32 LoRA linears that produce the SAME output -- the backward-data pass of the q | k | v
// projections of a self-attention (scripts/lib/original_unet.py:688-704): dX = dQ.Wq + dK.Wk + dV.Wv + the three LoRA terms.
// Before: three launches, each adding the previous dX in its epilogue (the first writes 2 B/element, the others read + write it);
// a 13 us launch of this size is ~75 % fixed cost (tools/experiments/qkv_bwd_bound.py: three chained launches 39.4 us, one
// launch with the same FLOPs 19.2 us at 16384x320x(3x320)).  Here the accumulators stay in registers across the groups; per
// group the kernel runs lora_gemm_kernel_w's K loop and LoRA up step (stage 0 of the drained ring holds Ts and the Bup panel while
// the next group's first tiles already fly into the other stages), T_g / Ts_g go to their own outputs, one epilogue stores the tile.
// Accumulation order differs from the chained launches (one fp32 accumulator instead of bf16 round trips between groups).
#pragma once
#include "aql_gemm.cuh"

namespace aqlkg {
using namespace aqlgemm;

constexpr int KG_LR = 32;
constexpr int KG_MAX = 3;

struct KGArgs {
  int ng;
  PlainLoader x[KG_MAX];     // [M][K_g]
  PlainLoader w[KG_MAX];     // [N][K_g]
  PlainLoader ad[KG_MAX];    // LoRA down [32][K_g]
  const bf16_t* bup[KG_MAX]; // LoRA up [N][32]
  bf16_t* T[KG_MAX];         // [M][32] out
  bf16_t* Ts[KG_MAX];
  int kt[KG_MAX];            // K tiles per group
  const bf16_t* S;           // [nsamples][32]
  int rps;
  int M, N, m_fast;
  EpiParams epi;             // C, ldc, residual, ldr (no bias / GEGLU in this form)
};

template <int BM, int BN, int WM, int WN, int NSTG>
__global__ __launch_bounds__(2 * NTHREADS) void lora_gemm_kernel_wk(const KGArgs a) {
  constexpr int FM = WM / 16, FN = WN / 16;
  constexpr int WAVES_N = BN / WN;
  static_assert((BM / WM) * (BN / WN) == 4, "4 compute wavefronts (+ 4 loader wavefronts) per workgroup");
  constexpr int FT = KG_LR / (16 * WAVES_N);
  static_assert(FT >= 1 && NSTG >= 3, "ring of >= 3 stages");
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, L_BYTES = KG_LR * 128;
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
  const int tiles_n = (a.N + BN - 1) / BN, tiles_m = (a.M + BM - 1) / BM;
  int tile_m, tile_n;
  if (a.m_fast) {
    tile_n = block_x / tiles_m;
    tile_m = block_x - tile_n * tiles_m;
  } else {
    tile_m = block_x / tiles_n;
    tile_n = block_x - tile_m * tiles_n;
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const bool t_writer = tile_n == 0;
  constexpr int NLD = BM / 32 + BN / 32 + 1;

  f32x4_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  constexpr int NBP = (BN * 4 + NTHREADS - 1) / NTHREADS;
  char* const sA = lds;
  char* const sB = lds + A_BYTES;

  // The two roles are two separate programs with the same barrier sequence per group -- P, one per K tile, B1 (ring drained),
  // B2 (Ts and the Bup panel are in stage 0), B3 (up step done) -- so that the stager state and the accumulators never share a
  // live range (sharing them spilled 868 bytes per lane in the 128-row form).
  // Ring placement: group 0 starts in stages 0 .. NSTG-2; every later group is pre-issued into stages 1 .. NSTG-1 BEFORE the up
  // step of its predecessor (which owns stage 0), so the ring refill flies under that up step instead of behind it.
  if (loader) {
    uint4 bup[KG_MAX][NBP];   // every group's Bup panel, fetched once up front
#pragma unroll
    for (int gi = 0; gi < KG_MAX; ++gi)
#pragma unroll
      for (int u = 0; u < NBP; ++u) {
        const int id = ltid + u * NTHREADS, row = id >> 2, c = id & 3;
        const int brow = n0 + row;
        const bool ok = (gi < a.ng) & (id < BN * 4) & (brow < a.N);
        bup[gi][u] = epi_mask4(*reinterpret_cast<const uint4*>(a.bup[gi < a.ng ? gi : 0] + (ok ? (long)brow * KG_LR + c * 8 : 0)), ok);
      }
    DmaStager<BM, PlainLoader> sa;
    DmaStager<BN, PlainLoader> sb;
    DmaStager<KG_LR, PlainLoader> sl;
    auto issue = [&](int stage) {
      char* st = lds + stage * STAGE;
      sa.dma(st, wave - 4);
      sb.dma(st + A_BYTES, wave - 4);
      sl.dma(st + A_BYTES + B_BYTES, wave - 4);
    };
    auto start_group = [&](int gi, int first_stage) {
      const int kt_end = a.kt[gi];
      sa.begin(a.x[gi], a.x[gi], false, m0, ltid, 0, kt_end, kt_end);
      sb.begin(a.w[gi], a.w[gi], false, n0, ltid, 0, kt_end, kt_end);
      sl.begin(a.ad[gi], a.ad[gi], false, 0, ltid, 0, kt_end, kt_end);
#pragma unroll
      for (int u = 0; u < NSTG - 1; ++u) issue(first_stage + u);
    };
    start_group(0, 0);
    for (int gi = 0; gi < a.ng; ++gi) {
      const int kt_end = a.kt[gi];
      const int s0 = gi == 0 ? 0 : 1;
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTG - 2) * NLD) : "memory");  // first tile landed (older loads too)
      __builtin_amdgcn_s_barrier();                                             // P
      int wr = (s0 + NSTG - 1) % NSTG;
      for (int kt = 0; kt < kt_end; ++kt) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTG - 3) * NLD) : "memory");  // tile kt+1 landed
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue(wr);
        wr = (wr + 1 == NSTG) ? 0 : wr + 1;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trailing zero-fill DMAs still write LDS
      __builtin_amdgcn_s_barrier();                      // B1
      asm volatile("" ::: "memory");
#pragma unroll
      for (int u = 0; u < NBP; ++u) {
        const int id = ltid + u * NTHREADS, row = id >> 2, c = id & 3;
        uint4 v = bup[0][u];
#pragma unroll
        for (int q = 1; q < KG_MAX; ++q)
          if (gi == q) v = bup[q][u];   // (uniform selects: a runtime-indexed register array would go to scratch)
        if (id < BN * 4) *reinterpret_cast<uint4*>(sB + lds_off(row, c)) = v;
      }
      if (gi + 1 < a.ng) start_group(gi + 1, 1);   // stages 1 .. NSTG-1 are free; stage 0 serves the up step
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the panel is in LDS (the DMAs of the next group stay in flight)
      __builtin_amdgcn_s_barrier();                        // B2
      __builtin_amdgcn_s_barrier();                        // B3
    }
  } else {
    uint2 srow[FM][FT];      // this lane's scale rows (the same for every group)
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const long m = (long)m0 + wm0 + i * 16 + (lane & 15);
#pragma unroll
      for (int t = 0; t < FT; ++t) {
        const bool ok = m < a.M;
        srow[i][t] = epi_mask2(*reinterpret_cast<const uint2*>(a.S + (ok ? (long)((uint32_t)m / (uint32_t)a.rps) * KG_LR + (wt0 + t) * 16 + (lane >> 4) * 4 : 0)), ok);
      }
    }
    const int arow = wm0 + (lane & 15), brow = wn0 + (lane & 15), lrow = wt0 * 16 + (lane & 15);
    const int ch0 = lane >> 4, ch1 = 4 + (lane >> 4);
    for (int gi = 0; gi < a.ng; ++gi) {
      const int kt_end = a.kt[gi];
      const int s0 = gi == 0 ? 0 : 1;
      f32x4_t tacc[FM][FT];
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int t = 0; t < FT; ++t) tacc[i][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      bf16x8_t fa0[FM], fb0[FN], fl0[FT], fa1[FM], fb1[FN], fl1[FT];
      __builtin_amdgcn_s_barrier();                       // P
      asm volatile("" ::: "memory");
      const char* f0 = lds + s0 * STAGE;
#pragma unroll
      for (int j = 0; j < FN; ++j) fb0[j] = *reinterpret_cast<const bf16x8_t*>(f0 + A_BYTES + lds_off(brow + j * 16, ch0));
#pragma unroll
      for (int t = 0; t < FT; ++t) fl0[t] = *reinterpret_cast<const bf16x8_t*>(f0 + A_BYTES + B_BYTES + lds_off(lrow + t * 16, ch0));
#pragma unroll
      for (int i = 0; i < FM; ++i) fa0[i] = *reinterpret_cast<const bf16x8_t*>(f0 + lds_off(arow + i * 16, ch0));
      int rd = s0;
      for (int kt = 0; kt < kt_end; ++kt) {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const char* cA = lds + rd * STAGE;
        const char* cB = cA + A_BYTES;
        const char* cL = cB + B_BYTES;
        rd = (rd + 1 == NSTG) ? 0 : rd + 1;
        const char* nA = lds + rd * STAGE;
        const char* nB = nA + A_BYTES;
        const char* nL = nB + B_BYTES;
#pragma unroll
        for (int j = 0; j < FN; ++j) fb1[j] = *reinterpret_cast<const bf16x8_t*>(cB + lds_off(brow + j * 16, ch1));
#pragma unroll
        for (int t = 0; t < FT; ++t) fl1[t] = *reinterpret_cast<const bf16x8_t*>(cL + lds_off(lrow + t * 16, ch1));
#pragma unroll
        for (int i = 0; i < FM; ++i) fa1[i] = *reinterpret_cast<const bf16x8_t*>(cA + lds_off(arow + i * 16, ch1));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < FM; ++i) {
#pragma unroll
          for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb0[j], fa0[i], acc[i][j], 0, 0, 0);
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
#pragma unroll
          for (int t = 0; t < FT; ++t) tacc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fl1[t], fa1[i], tacc[i][t], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                       // B1: nobody reads the ring any more, its trailing DMAs have landed
      asm volatile("" ::: "memory");
      // ---- this group's LoRA up step: Ts -> stage-0 A tile (the loaders put the Bup panel into the stage-0 B tile), one k-step
      bf16_t* const Tg = a.T[gi];
      bf16_t* const Tsg = a.Ts[gi];
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int row = wm0 + i * 16 + (lane & 15);
        const long m = (long)m0 + row;
        const bool ok = m < a.M;
#pragma unroll
        for (int t = 0; t < FT; ++t) {
          const int r = (wt0 + t) * 16 + (lane >> 4) * 4;
          const uint2 tv = make_uint2(pack_bf16x2(tacc[i][t][0], tacc[i][t][1]), pack_bf16x2(tacc[i][t][2], tacc[i][t][3]));
          const uint2 sv = srow[i][t];
          const uint2 ts = make_uint2(pack_bf16x2(bf16lo(tv.x) * bf16lo(sv.x), bf16hi(tv.x) * bf16hi(sv.x)),
                                      pack_bf16x2(bf16lo(tv.y) * bf16lo(sv.y), bf16hi(tv.y) * bf16hi(sv.y)));
          *reinterpret_cast<uint2*>(sA + lds_off(row, r >> 3) + (r & 7) * 2) = ts;
          if (ok && t_writer) {
            *reinterpret_cast<uint2*>(Tg + m * KG_LR + r) = tv;
            *reinterpret_cast<uint2*>(Tsg + m * KG_LR + r) = ts;
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                       // B2
      asm volatile("" ::: "memory");
      {
        bf16x8_t fa[FM], fb[FN];
        const int chunk = lane >> 4;
#pragma unroll
        for (int i = 0; i < FM; ++i) fa[i] = *reinterpret_cast<const bf16x8_t*>(sA + lds_off(wm0 + i * 16 + (lane & 15), chunk));
#pragma unroll
        for (int j = 0; j < FN; ++j) fb[j] = *reinterpret_cast<const bf16x8_t*>(sB + lds_off(wn0 + j * 16 + (lane & 15), chunk));
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                       // B3
      asm volatile("" ::: "memory");
    }
  }
  __syncthreads();

  // ---- epilogue: the bf16 tile through LDS, residual added in bf16 (as lora_gemm_kernel_w)
  if (!loader) {
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int row = wm0 + i * 16 + (lane & 15);
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int col = wn0 + j * 16 + (lane >> 4) * 4;
        *reinterpret_cast<uint2*>(lds + row * C_PITCH + col * 2) =
            make_uint2(pack_bf16x2(acc[i][j][0], acc[i][j][1]), pack_bf16x2(acc[i][j][2], acc[i][j][3]));
      }
    }
  }
  __syncthreads();
  epi_store_tile<BM, BN, C_PITCH, 2 * NTHREADS>(lds, m0, n0, a.M, a.N, a.epi, tid);
}

template <int BM, int BN, int WM, int WN, int NSTG>
inline void launch_wk(const KGArgs& a, hipStream_t stream) {
  dim3 grid(aql_cdiv(a.M, BM) * aql_cdiv(a.N, BN));
  hipLaunchKernelGGL((lora_gemm_kernel_wk<BM, BN, WM, WN, NSTG>), grid, dim3(2 * NTHREADS), 0, stream, a);
}

}  // namespace aqlkg
