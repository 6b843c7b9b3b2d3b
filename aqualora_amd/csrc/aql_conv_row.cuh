// Row-tile form of the 12-wave 3x3 convolution kernel (stride 1, pad 1, 64-pixel-wide maps, Cin % 64 == 0): the forward
// convolutions of the 64x64 level of the U-Net (scripts/lib/original_unet.py ResnetBlock2D conv1 / conv2 as reached from the
// PPFT step, train/ppft_train.py:1026-1035).
//
// Why: a per-K-tile timeline of gemm_kernel_w<256,160,...,3,8> (tools/trace_gemm_w256.py, MI355X) shows the K loop at 1960 cycles
// per tile for 1304 cycles of MFMA per SIMD, and the four loader wavefronts busy the whole time: 13 LDS-DMA instructions per
// loader thread per tile at ~100 cycles each -- the CU takes ~40 B/clk of global -> LDS traffic, and an implicit-GEMM tile
// of 256 pixels x 64 channels is fetched again for every one of the 9 taps.  Here the workgroup's output tile is 4 whole image
// rows; for one (kh, 64-channel slab) it stages the 4 x 66 input pixels ONCE (two zero columns of padding included) and the
// three kw taps read their A fragments from that tile at a pixel offset of kw: per three K tiles 34 + 3 x 20 KB of DMA instead
// of 3 x 53 KB, 8 (not 13) DMA instructions per loader thread per tile.  K order: kh outer, channel slab, kw inner (the
// fp32 accumulation order differs from the tap-major kernels; same products).
//   LDS: two A tiles (9 x 32 pixel rows x 128 B = 36 KB each) + a 4-stage ring of 160 x 64 weight tiles (20 KB each) = 152 KB.
//   Loaders, per K tile t = 3 g + kw: [wait] [barrier] [A tile of group g+1: 5 instructions at kw = 0, 4 at kw = 1] [W tile t+3]
//   with counted vmcnt waits of 10 / 15 / 19 outstanding loads (two weight tiles always in flight; derivation in DESIGN.md,
//   section 4), compute wavefronts: [barrier] 2 x ([9 fragment reads] [20 MFMAs]).
#pragma once
#include "aql_gemm.cuh"
#include <stdlib.h>

namespace aqlconvrow {
using namespace aqlgemm;

constexpr int CR_BN = 160, CR_WM = 64, CR_WN = 80;
constexpr int WST = CR_BN * 128;               // 20,480 B per weight tile
constexpr int NSTW = 4;
constexpr int NW = CR_BN / 32;                 // DMA instructions per loader thread per weight tile

// Swizzle of the A tile.  The tap kw reads 16 consecutive pixel rows starting at ANY row (r (RW + 2) + 16 i + kw), and the (row >> 1) & 7
// swizzle of the weight tiles (lds_off) is conflict-free for ds_read_b128 only when that start is a multiple of 4: two of the three
// taps paid 8 LDS cycles per fragment instead of 4 (SQ_LDS_BANK_CONFLICT 2.21 M of 9.09 M LDS cycles per launch against 0.25 M on the
// implicit-GEMM kernel, profiles/r02_pmc_conv_row_lds.txt).  ((row >> 1) & 3) << 1 leaves bit 0 of the chunk to the lane group and is
// conflict-free for every start row (checked exhaustively, tools/experiments/lds_swizzle_check.py).
__device__ __forceinline__ int a_swz(int row) { return ((row >> 1) & 3) << 1; }
__device__ __forceinline__ int a_off(int row, int chunk) { return row * 128 + ((chunk ^ a_swz(row)) << 4); }

// What the kernel needs of the convolution (forward: ConvFwdLoader, backward-data: ConvBwdLoader with the taps flipped)
struct RowArgs {
  const bf16_t* x;      // [B][H][RW][C] channels-last input (forward) / output gradient (backward-data)
  int H, C;             // image rows, contraction channels
  PlainLoader w;        // [N][9 C] weights, column (kh*3+kw)*C + c
  int M, N, m_fast;
  int splits;           // grid.z: the (kh, slab) groups are divided evenly; > 1 only with the fp32 slab epilogue
  EpiParams epi;
};

// RW = image width, TROWS = image rows per output tile (BM = RW * TROWS = 256: 8 compute wavefronts, 128: 4), FLIP = backward-data
// (tap (kh, kw) reads pixel (h + 1 - kh, w + 1 - kw) of dY; forward: (h + kh - 1, w + kw - 1)).
template <int RW, int TROWS, bool FLIP>
struct RowCfg {
  static constexpr int BM = RW * TROWS;
  static constexpr int NCW = BM / 64 * 2;
  static constexpr int THREADS = (NCW + 4) * 64;
  static constexpr int APX = TROWS * (RW + 2);
  static constexpr int A_INSTR = (APX + 31) / 32;
  static constexpr int P1 = (A_INSTR + 1) / 2, P2 = A_INSTR - P1;   // A instructions issued at kw = 0 / kw = 1
  static constexpr int ABUF = A_INSTR * 32 * 128;
};

// BN = output columns per tile: 160 (the U-Net: every channel count is a multiple of 160) or 128 (the VAE: 128 / 256 / 512 channels;
// on 160-wide tiles a fifth of the MFMAs and weight loads would be padding).  Everything below derives from it.
// WI = image width when a tile is a PART of an image row (WI = 512 on RW = 256-pixel tiles, TROWS = 1): the tile's two halo columns are
// then the neighbouring half's pixels, not padding.
template <int RW, int TROWS, bool FLIP, bool SLAB, int BN = 160, int WI = RW>
__global__ __launch_bounds__((RowCfg<RW, TROWS, FLIP>::THREADS)) void conv_row_kernel(const RowArgs g) {
  static_assert(WI == RW || (TROWS == 1 && WI % RW == 0), "a partial-row tile is one image row high");
  using CFG = RowCfg<RW, TROWS, FLIP>;
  constexpr int CR_BN = BN, CR_WN = BN / 2, WST = BN * 128, NW = BN / 32;   // shadow the 160-wide defaults of the namespace
  constexpr int CR_BM = CFG::BM, NCW = CFG::NCW, CR_THREADS = CFG::THREADS, APX = CFG::APX, A_INSTR = CFG::A_INSTR, ABUF = CFG::ABUF;
  constexpr int P1 = CFG::P1;
  constexpr int FM = CR_WM / 16, FN = CR_WN / 16;
  constexpr int WAVES_N = CR_BN / CR_WN;
  constexpr int C_PITCH = (CR_BN + 8) * 2;
  constexpr int RING = 2 * ABUF + NSTW * WST;
  constexpr int LDS_BYTES = RING > CR_BM * C_PITCH ? RING : CR_BM * C_PITCH;
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
  __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
  char* const abuf = lds;
  char* const wring = lds + 2 * ABUF;

  const int nblk = gridDim.x, bid = blockIdx.x;
  const int qq = nblk >> 3, rr = nblk & 7, xcd = bid & 7;
  const int block_x = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wave >= NCW;
  const int tiles_n = (g.N + CR_BN - 1) / CR_BN, tiles_m = g.M / CR_BM;
  int tile_m, tile_n;
  if (g.m_fast) {
    tile_n = block_x / tiles_m;
    tile_m = block_x - tile_n * tiles_m;
  } else {
    tile_m = block_x / tiles_n;
    tile_n = block_x - tile_m * tiles_n;
  }
  const int m0 = tile_m * CR_BM, n0 = tile_n * CR_BN;
  const int H = g.H, Cin = g.C;
  const int nslab = Cin >> 6;
  const int NGall = 3 * nslab;                    // (kh, slab) groups; 3 K tiles each
  const int ga = (int)(((long)NGall * blockIdx.z) / g.splits);          // this split's groups [ga, NG)
  const int NG = (int)(((long)NGall * (blockIdx.z + 1)) / g.splits);
  const int T = 3 * NG;

  f32x4_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  uint2 biasr[FN];
#ifdef AQL_TRACE_W
  long* trw = (block_x == 0 && lane == 0 && g.epi.Cf != nullptr) ? reinterpret_cast<long*>(g.epi.Cf) + wave * 1024 : nullptr;
#define CRT(i, slot) do { if (trw && (i) < 255) trw[(i) * 4 + (slot)] = clock64(); } while (0)
#else
#define CRT(i, slot) do { } while (0)
#endif

  if (loader) {
    const int lw = wave - NCW;
    const int ltid = tid - NCW * 64;
    const int b = m0 / (H * WI), h0 = (m0 - b * H * WI) / WI, x0 = m0 - (b * H + h0) * WI;   // x0 = 0 unless WI > RW
    __amdgpu_buffer_rsrc_t rsx = make_rsrc(g.x), rsw = make_rsrc(g.w.base);
    // A tile: instruction j stages pixel rows 32 j .. 32 j + 31; this lane: row 32 j + (ltid >> 3), 16-byte slot ltid & 7
    uint32_t abase[A_INSTR];
    int arh[A_INSTR];           // input image row of the pixel for kh = 0 (h0 + r - 1), or a value that never passes the test
#pragma unroll
    for (int j = 0; j < A_INSTR; ++j) {
      const int p = 32 * j + (ltid >> 3);
      const int r = p / (RW + 2), wc = p - r * (RW + 2) - 1;
      const int kc = ((ltid & 7) ^ a_swz(p)) * 8;
      const bool okc = (p < APX) & (x0 + wc >= 0) & (x0 + wc < WI);
      arh[j] = okc ? h0 + r - 1 : -(1 << 20);
      abase[j] = (uint32_t)((((long)b * H + (h0 + r - 1)) * WI + x0 + wc) * Cin + kc) * 2u;
    }
    uint32_t wv[CR_BN / 32];
#pragma unroll
    for (int i = 0; i < CR_BN / 32; ++i) {
      const int row = n0 + (ltid >> 3) + 32 * i;
      const int kc = ((ltid & 7) ^ ((ltid >> 4) & 7)) * 8;
      wv[i] = row < g.w.rows ? (uint32_t)row * (uint32_t)(g.w.ld * 2) + kc * 2 : OOB_ROW;
    }
    auto issueA = [&](int gi, int lo, int hi) {   // group gi = kh * nslab + c; instructions [lo, hi)
      const int kh = gi / nslab, c = gi - kh * nslab;
      const int khe = FLIP ? 2 - kh : kh;          // image-row offset of the tap, plus one
      const uint32_t add = (uint32_t)((khe * WI * Cin + c * 64) * 2);
      char* dst = abuf + (gi & 1) * ABUF;
      const bool live = gi < NG;
#pragma unroll
      for (int j = 0; j < A_INSTR; ++j) {
        if (j < lo || j >= hi) continue;
        const bool ok = live & ((unsigned)(arh[j] + khe) < (unsigned)H);
        dma16(rsx, dst + (32 * j + 8 * lw) * 128, ok ? abase[j] + add : OOB_ROW, 0);
      }
    };
    auto issueW = [&](int t) {   // K tile t = 3 (kh * nslab + c) + kw  ->  weight columns (kh * 3 + kw) * Cin + 64 c
      const int gi = t / 3, kw = t - gi * 3;
      const int kh = gi / nslab, c = gi - kh * nslab;
      const uint32_t soff = (uint32_t)(((kh * 3 + kw) * Cin + c * 64) * 2);
      char* dst = wring + (t & (NSTW - 1)) * WST;
      const bool live = t < T;
#pragma unroll
      for (int i = 0; i < CR_BN / 32; ++i) dma16(rsw, dst + (32 * i + 8 * lw) * 128, live ? wv[i] : OOB_ROW, live ? soff : 0);
    };
    issueA(ga, 0, A_INSTR);
    issueW(3 * ga);
    issueW(3 * ga + 1);
    issueW(3 * ga + 2);
    for (int gi = ga; gi < NG; ++gi) {
      const int t = 3 * gi;
      CRT(t, 0);
      // barrier(t) certifies that tile t (and, at t = 3g, A tile g) has landed and that nobody reads tile t-1 any more.  Loads that
      // may still be in flight at each wait, in issue order (A part before W tile in an iteration):
      //   t = 3g  : W(t+1), W(t+2)                                              -> 2 NW
      //   t = 3g+1: W(t+1), A(g+1) part 1, W(t+2)                               -> P1 + 2 NW
      //   t = 3g+2: A(g+1) part 1, W(t+1), A(g+1) part 2, W(t+2)                -> A_INSTR + 2 NW
      // (a register-double-buffered form that fetched the first fragments of tile t+1 during tile t measures the same in sustained
      // runs, 48-49 us at 8 x 64 x 64 x 320 -> 320, and needs 164 instead of 123 VGPRs; with 14 of them spilled it was 17 % slower:
      // profiles/r02_ab_conv_row_prefetch.txt.  tools/check_spills.py lists every kernel's scratch use.)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NW) : "memory");
      CRT(t, 1);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      CRT(t, 2);
      issueA(gi + 1, 0, P1);
      issueW(t + 3);
      CRT(t, 3);
      CRT(t + 1, 0);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P1 + 2 * NW) : "memory");
      CRT(t + 1, 1);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      CRT(t + 1, 2);
      issueA(gi + 1, P1, A_INSTR);
      issueW(t + 4);
      CRT(t + 1, 3);
      CRT(t + 2, 0);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_INSTR + 2 * NW) : "memory");
      CRT(t + 2, 1);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      CRT(t + 2, 2);
      issueW(t + 5);
      CRT(t + 2, 3);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // trailing zero-fill DMAs still write LDS
  } else {
    const int wn0 = (wave % WAVES_N) * CR_WN;
    // this wavefront's 64 pixels: tile pixels (wave / WAVES_N) * 64 ..; a 16-pixel fragment lies inside one image row (RW >= 16):
    // A-tile pixel row of pixel (r, w) for tap kw = r (RW + 2) + w + kw  (backward-data: + 2 - kw)
    const int pt0 = (wave / WAVES_N) * CR_WM + (lane & 15);
    const int prow0 = (pt0 / RW) * (RW + 2) + pt0 % RW;
    // fragment i starts 16 i pixels further: a compile-time offset in A-tile pixel rows (RW divides 64)
    auto prow = [&](int i) { return prow0 + (i * 16 / RW) * (RW + 2) + (i * 16) % RW; };
    const int brow = wn0 + (lane & 15);
    const int g4 = lane >> 4;
    for (int gi = ga; gi < NG; ++gi) {
      const char* sA = abuf + (gi & 1) * ABUF;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int t = 3 * gi + kw;
        CRT(t, 0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        CRT(t, 1);
        const char* sB = wring + (t & (NSTW - 1)) * WST;
        const int po = FLIP ? 2 - kw : kw;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const int chunk = ks * 4 + g4;
          bf16x8_t fa[FM], fb[FN];
#pragma unroll
          for (int i = 0; i < FM; ++i) fa[i] = *reinterpret_cast<const bf16x8_t*>(sA + a_off(prow(i) + po, chunk));
#pragma unroll
          for (int j = 0; j < FN; ++j) fb[j] = *reinterpret_cast<const bf16x8_t*>(sB + lds_off(brow + j * 16, chunk));
#pragma unroll
          for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
        CRT(t, 2);
      }
    }
    // (after the loop: five batched loads, one round trip per workgroup; held across the loop they spilled)
    if constexpr (!SLAB) epi_load_bias<FN>(biasr, g.epi.bias, g.w.base, n0, wn0, lane, g.N, 0, CR_BN / 2);
  }
  if constexpr (SLAB) {   // split K: this split's fp32 partial tile, finished by splitk_finalize_kernel
    if (!loader) {
      const int wm0 = (wave / WAVES_N) * CR_WM, wn0 = (wave % WAVES_N) * CR_WN;
      float* out = g.epi.Cf + (long)blockIdx.z * g.M * g.epi.ldcf;
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int m = m0 + wm0 + i * 16 + (lane & 15);
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const int n = n0 + wn0 + j * 16 + (lane >> 4) * 4;
          if (m >= g.M || n >= g.N) continue;
          *reinterpret_cast<float4*>(out + (long)m * g.epi.ldcf + n) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        }
      }
    }
    return;
  }
  __syncthreads();

  // ---- epilogue (as gemm_body_w): bias in fp32, bf16 tile through LDS, row bias / residual in the store loop
  const EpiParams& ep = g.epi;
  if (!loader) {
    const int wm0 = (wave / WAVES_N) * CR_WM, wn0 = (wave % WAVES_N) * CR_WN;
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
  epi_store_tile<CR_BM, CR_BN, C_PITCH, CR_THREADS>(lds, m0, n0, g.M, g.N, ep, tid);
}

template <int RW, int TROWS, bool FLIP, bool SLAB, int BN = 160, int WI = RW>
inline void launch_conv_row(const RowArgs& a, hipStream_t stream) {
  using CFG = RowCfg<RW, TROWS, FLIP>;
  dim3 grid((a.M / CFG::BM) * aql_cdiv(a.N, BN), 1, a.splits);
  hipLaunchKernelGGL((conv_row_kernel<RW, TROWS, FLIP, SLAB, BN, WI>), grid, dim3(CFG::THREADS), 0, stream, a);
}

// bm = 256 or 128 (the tile height the picker chose).  Returns false when the convolution does not fit a row tile.
template <class LA, int EPI>
inline bool try_conv_row(const GemmArgs<LA, PlainLoader>& g, int bm, hipStream_t stream) {
  constexpr bool FLIP = std::is_same<LA, ConvBwdLoader>::value;
  constexpr bool SLAB = EPI == EPI_SLAB;
  const auto& l = g.a0;
  int H, W, C;
  if constexpr (FLIP) {
    if (l.stride != 1 || l.Hin != l.Hout || l.Win != l.Wout) return false;
    H = l.Hin, W = l.Win, C = l.Cout;
  } else {
    if (l.stride != 1 || l.ups != 0 || l.pad != 1 || l.Hin != l.Hout || l.Win != l.Wout) return false;
    H = l.Hin, W = l.Win, C = l.Cin;
  }
  if (C % 64 != 0 || g.N % 8 != 0 || g.epi.geglu_F != 0 || g.ktiles1 != 0) return false;
  static const int slab_ok = AQL_TUNE_INT("AQL_CONV_ROW_SLAB", 1);   // A/B hook
  if ((g.splits != 1) != SLAB || g.splits > 3 * (C / 64) || (SLAB && !slab_ok)) return false;
  static const int row32x8 = AQL_TUNE_INT("AQL_CONV_ROW_32X8", 1);   // A/B hook
  RowArgs a;
  a.x = l.base, a.H = H, a.C = C, a.w = g.b0, a.M = g.M, a.N = g.N, a.m_fast = g.m_fast, a.splits = g.splits, a.epi = g.epi;
  const bool n128 = g.N % 128 == 0 && g.N % 160 != 0;    // the VAE's channel counts
  if (bm == 256 && W == 512 && n128) launch_conv_row<256, 1, FLIP, SLAB, 128, 512>(a, stream);   // the VAE's 512-pixel rows as two half-row tiles
  else if (bm == 256 && W == 256 && n128) launch_conv_row<256, 1, FLIP, SLAB, 128>(a, stream);    // the VAE's 256- and 128-pixel-wide maps
  else if (bm == 256 && W == 256) launch_conv_row<256, 1, FLIP, SLAB>(a, stream);
  else if (bm == 256 && W == 128 && H % 2 == 0 && n128) launch_conv_row<128, 2, FLIP, SLAB, 128>(a, stream);
  else if (bm == 256 && W == 128 && H % 2 == 0) launch_conv_row<128, 2, FLIP, SLAB>(a, stream);
  else if (bm == 256 && W == 64 && H % 4 == 0 && n128) launch_conv_row<64, 4, FLIP, SLAB, 128>(a, stream);
  else if (bm == 256 && W == 64 && H % 4 == 0) launch_conv_row<64, 4, FLIP, SLAB>(a, stream);
  else if (bm == 256 && W == 32 && H % 8 == 0 && row32x8) launch_conv_row<32, 8, FLIP, SLAB>(a, stream);
  else if (bm == 128 && W == 64 && H % 2 == 0) launch_conv_row<64, 2, FLIP, SLAB>(a, stream);
  else if (bm == 128 && W == 32 && H % 4 == 0) launch_conv_row<32, 4, FLIP, SLAB>(a, stream);
  else if (bm == 128 && W == 16 && H % 8 == 0) launch_conv_row<16, 8, FLIP, SLAB>(a, stream);
  else return false;
  return true;
}

}  // namespace aqlconvrow
