// Token-reduction ("TN") GEMM for the wide-rank LoRA weight gradients:  C[P,Q] (fp32) += alpha * U[M,P]^T V[M,Q]
// (dA = dT^T X and dB = dY^T Ts of utils/lora_modules.py:13-19 under autograd, at rank 320 real GEMMs of
// 2*320*C*tokens FLOP each).  Both operands are stored token-major, i.e. the contraction index is the ROW index, which
// is the wrong way round for an MFMA fragment (8 consecutive contraction elements per lane).  Instead of transposing
// both operands in HBM first (the previous path: 2 transpose launches + NT GEMM + split-K accumulate per gradient,
// 17 % of the rank-320 step), the 64-token tiles are staged row-major exactly as they lie in memory and the fragments
// are gathered with the LDS transpose read ds_read_b64_tr_b16 -- the same gather as the P.V product of the attention
// kernels (aql_attn.hip, t_product).  Both operands use the identical gather, so the k-slot permutation it implies
// (slot (g,e) of step s <-> token 32 s + 16 (e>>2) + 4 g + (e&3)) cancels.
//
// Workgroup: 256 threads = 2x2 wavefronts, tile 128 (P) x 128 (Q), 64x64 per wavefront (16 accumulators of 16x16);
// LDS 2 x 16 KB, next tile prefetched into registers under the MFMAs; the token range is split over grid.y and the
// partial tiles are accumulated with fp32 atomics (the weight gradients are order-nondeterministic in the last bits
// already).
#include "aql_common.h"
#include <stddef.h>

namespace {

constexpr int TK = 64;      // tokens per stage
constexpr int BT = 128;     // tile width, both sides
constexpr int PITCH = 256;  // bytes per LDS row

// conflict-free for the 8-row x 32-byte transpose gathers (same image as aql_attn.hip's row tiles)
template <int PB>
__device__ __forceinline__ int tile_off_p(int row, int chunk) { return row * PB + ((chunk ^ ((row & 7) << 1)) << 4); }
__device__ __forceinline__ int tile_off(int row, int chunk) { return tile_off_p<PITCH>(row, chunk); }
// 160-column tiles (round 6): 20 live chunks per row; the XOR moves a chunk by up to 14 places, so the row is 32 chunks (512 bytes)
// long -- the same bank picture as the 256-byte rows (a row starts on bank 0 either way)
constexpr int PITCH160 = 512;
template <int W>
struct PitchOf { static constexpr int value = W == 160 ? PITCH160 : PITCH; };

__device__ __forceinline__ uint4 mask4(const uint4& v, bool ok) {
  const uint32_t m = 0u - (uint32_t)ok;
  return make_uint4(v.x & m, v.y & m, v.z & m, v.w & m);
}

template <int W>  // tile width in columns: 128 (16 chunks per row, 4 slots per thread), 160 (20 chunks, 5 slots) or 32 (4 chunks, 1 slot)
struct TileStager {
  static constexpr int CPRW = W / 8, SLOTS = TK * CPRW / 256, PB = PitchOf<W>::value;
  long p[SLOTS];   // element offset from g of the slot in the CURRENT tile
  uint4 v[SLOTS];
  int off[SLOTS];  // LDS byte offset, -1: column past the operand's width (zeroed once)
  const bf16_t* g;
  long ld;

  __device__ __forceinline__ void init(char* lds, const bf16_t* g_, long ld_, long m0, int col0, int width, int tid) {
    g = g_;
    ld = ld_;
#pragma unroll
    for (int it = 0; it < SLOTS; ++it) {
      const int id = tid + it * 256;
      const int row = id / CPRW, c = id - row * CPRW;
      const bool live = col0 + c * 8 < width;
      off[it] = live ? tile_off_p<PB>(row, c) : -1;
      p[it] = live ? (m0 + row) * ld + col0 + c * 8 : 0;
      if (!live) *reinterpret_cast<uint4*>(lds + tile_off_p<PB>(row, c)) = make_uint4(0u, 0u, 0u, 0u);
    }
  }
  // full tile: unconditional loads; tail tile (rows past M must read as ZERO: they enter the sum): clamped + masked
  __device__ __forceinline__ void fetch(long m0, long M, int tid) {
    if (m0 + TK <= M) {
#pragma unroll
      for (int it = 0; it < SLOTS; ++it) v[it] = *reinterpret_cast<const uint4*>(g + p[it]);
    } else {
#pragma unroll
      for (int it = 0; it < SLOTS; ++it) {
        const int row = (tid + it * 256) / CPRW;
        const bool ok = (m0 + row < M) & (off[it] >= 0);
        const uint4 x = *reinterpret_cast<const uint4*>(g + (ok ? p[it] : 0));
        v[it] = mask4(x, ok);
      }
    }
  }
  __device__ __forceinline__ void commit(char* lds) const {
#pragma unroll
    for (int it = 0; it < SLOTS; ++it)
      if (off[it] >= 0) *reinterpret_cast<uint4*>(lds + off[it]) = v[it];
  }
  __device__ __forceinline__ void advance() {
#pragma unroll
    for (int it = 0; it < SLOTS; ++it) p[it] += (long)TK * ld;  // dead slots stay at offset 0 + k*TK*ld: unused
  }
};

typedef short v4s_t __attribute__((ext_vector_type(4)));

// 16 columns [col0, col0+16) of the row-major tile, tokens 32*s2 .. 32*s2+31, as an MFMA operand (see header)
template <int PB = PITCH>
__device__ __forceinline__ bf16x8_t tr_frag(const char* tile, int s2, int col0, int lane) {
  const int p = lane & 15, g = lane >> 4;
  const int row = s2 * 32 + g * 4 + (p >> 2);  // rows row and row+16 share (row & 7): same swizzle
  const char* base = tile + tile_off_p<PB>(row, (col0 >> 3) + ((p & 3) >> 1)) + (p & 1) * 8;
  const v4s_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s_t __attribute__((address_space(3)))*)(base));
  const v4s_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s_t __attribute__((address_space(3)))*)(base + 16 * PB));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

struct TnArgs {
  const bf16_t *U, *V;
  long ldu, ldv, M;
  int P, Q;
  float alpha;
  float* C;
  long ldc;
  int tiles_q, tiles_per_split;
  int narrow;  // 1: Q <= 32 -> 128 x 32 tiles (LoRA rank <= 32); the caller swaps operands so the narrow side is Q
  int trans;   // 1: the result is written transposed, element (p, q) -> C[q * ldc + p]
};

// accumulator fragments per wavefront: 64 x 64 (tile 128 x 128), 64 x 80 (128 x 160) or 32 x 32 (128 x 32)
template <int BQ> struct FragsOf { static constexpr int FM = BQ == 32 ? 2 : 4, FN = BQ == 128 ? 4 : (BQ == 160 ? 5 : 2); };

template <int BQ, bool TRANS>
__device__ __forceinline__ void tn_tr_store(const TnArgs& a, const f32x4_t (&acc)[FragsOf<BQ>::FM][FragsOf<BQ>::FN], int p0, int q0,
                                            int wm0, int wn0, int lane) {
  constexpr int FM = FragsOf<BQ>::FM, FN = FragsOf<BQ>::FN;
  if (!TRANS) {
    // acc[i][j][e]: row P = p0 + wm0 + 16 i + 4 (lane>>4) + e, column Q = q0 + wn0 + 16 j + (lane & 15)
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int pr = p0 + wm0 + i * 16 + (lane >> 4) * 4 + e;
        if (pr >= a.P) continue;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const int qc = q0 + wn0 + j * 16 + (lane & 15);
          if (qc < a.Q) atomicAdd(a.C + (long)pr * a.ldc + qc, a.alpha * acc[i][j][e]);
        }
      }
  } else {
    // operands swapped: acc[i][j][e] is (Q = q0 + wn0 + 16 j + 4 (lane>>4) + e, P = p0 + wm0 + 16 i + (lane & 15)) -> C[Q][P]
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int qr = q0 + wn0 + j * 16 + (lane >> 4) * 4 + e;
        if (qr >= a.Q) continue;
#pragma unroll
        for (int i = 0; i < FM; ++i) {
          const int pc = p0 + wm0 + i * 16 + (lane & 15);
          if (pc < a.P) atomicAdd(a.C + (long)qr * a.ldc + pc, a.alpha * acc[i][j][e]);
        }
      }
  }
}

// BQ = 128: 2x2 wavefronts of 64x64.  BQ = 32 (rank <= 32 gradients): 4x1 wavefronts of 32x32, the narrow operand's tile
// uses 64 of its 256-byte LDS rows (same swizzle, so the gathers stay conflict-free).  TRANS swaps the MFMA operand roles
// so that consecutive lanes still hit consecutive addresses of the transposed output.
template <int BQ, bool TRANS>
__device__ __forceinline__ void tn_tr_body(const TnArgs& a, const int tile, const int split, char* sU, char* sV) {
  constexpr int FM = FragsOf<BQ>::FM, FN = FragsOf<BQ>::FN, PBV = PitchOf<BQ>::value;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tp = tile / a.tiles_q, tq = tile - tp * a.tiles_q;
  const int p0 = tp * BT, q0 = tq * BQ;
  const int wm0 = BQ == 32 ? wave * 32 : (wave >> 1) * 64, wn0 = BQ == 32 ? 0 : (wave & 1) * (BQ / 2);
  const long m_lo = (long)split * a.tiles_per_split * TK;
  long m_hi = m_lo + (long)a.tiles_per_split * TK;
  if (m_hi > a.M) m_hi = a.M;
  if (m_lo >= m_hi) return;

  f32x4_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  TileStager<BT> su;
  TileStager<BQ> sv;
  su.init(sU, a.U, a.ldu, m_lo, p0, a.P, tid);
  sv.init(sV, a.V, a.ldv, m_lo, q0, a.Q, tid);
  su.fetch(m_lo, a.M, tid);
  sv.fetch(m_lo, a.M, tid);
  for (long m = m_lo; m < m_hi; m += TK) {
    __syncthreads();
    su.commit(sU);
    sv.commit(sV);
    __syncthreads();
    if (m + TK < m_hi) {  // next tile's loads fly under this tile's MFMAs
      su.advance();
      sv.advance();
      su.fetch(m + TK, a.M, tid);
      sv.fetch(m + TK, a.M, tid);
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      bf16x8_t fa[FM], fb[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) fa[i] = tr_frag(sU, s2, wm0 + i * 16, lane);
#pragma unroll
      for (int j = 0; j < FN; ++j) fb[j] = tr_frag<PBV>(sV, s2, wn0 + j * 16, lane);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          acc[i][j] = TRANS ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0)
                            : __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
  }
  tn_tr_store<BQ, TRANS>(a, acc, p0, q0, wm0, wn0, lane);
}

// ---- LDS-DMA ring form.  The register-prefetch body above keeps ONE token tile per workgroup in flight (2 x 32 KB per CU): by
// Little's law ~16 B/clk/CU at the ~4k-cycle latency the loads see under load, which is what it measures (447 TFLOP/s on the
// rank-320 gradients).  Here the tiles go global -> LDS directly (buffer_load ... lds, 1 KB per wave instruction, the XOR
// swizzle of tile_off applied on the SOURCE side) into a ring of NST stages: NST-1 tiles in flight, no commit phase, ONE
// barrier per tile.  Rows past M and columns past the operand's width are fetched with an out-of-range offset (zero fill).
constexpr uint32_t TNTR_OOB = 0x80000000u;
constexpr uint32_t TNTR_BUF = 0x40000000u;   // operands must span < 1 GiB from their base (tn_tr_fill checks)
constexpr int TNTR_NI = 8;                   // DMA instructions per wavefront per stage: 4 (U tile) + 4 (V tile)

template <int BQ, bool TRANS, int NST>
__device__ __forceinline__ void tn_tr_body_dma(const TnArgs& a, const int tile, const int split, char* lds) {
  constexpr int FM = FragsOf<BQ>::FM, FN = FragsOf<BQ>::FN;
  constexpr int TILE_B = TK * PITCH, STAGE = 2 * TILE_B;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tp = tile / a.tiles_q, tq = tile - tp * a.tiles_q;
  const int p0 = tp * BT, q0 = tq * BQ;
  const int wm0 = BQ == 128 ? (wave >> 1) * 64 : wave * 32, wn0 = BQ == 128 ? (wave & 1) * 64 : 0;
  const long m_lo = (long)split * a.tiles_per_split * TK;
  long m_hi = m_lo + (long)a.tiles_per_split * TK;
  if (m_hi > a.M) m_hi = a.M;
  if (m_lo >= m_hi) return;
  const int ntiles = (int)((m_hi - m_lo + TK - 1) / TK);

  f32x4_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  __amdgpu_buffer_rsrc_t rsu = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.U), 0, TNTR_BUF, 0x00020000);
  __amdgpu_buffer_rsrc_t rsv = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.V), 0, TNTR_BUF, 0x00020000);
  // instruction i of this wavefront fills LDS rows 16 i + 4 wave + (lane >> 4) of a tile; lane slot (lane & 15) holds source chunk
  // slot ^ ((row & 7) << 1)
  uint32_t vu[4], vv[4];
  int rw[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 16 * i + 4 * wave + (lane >> 4);
    const int c = (lane & 15) ^ ((row & 7) << 1);
    rw[i] = row;
    vu[i] = (p0 + c * 8 < a.P) ? (uint32_t)(((long)row * a.ldu + p0 + c * 8) * 2) : TNTR_OOB;
    vv[i] = (c * 8 < BQ && q0 + c * 8 < a.Q) ? (uint32_t)(((long)row * a.ldv + q0 + c * 8) * 2) : TNTR_OOB;
  }
  // piece idx of tile t: 0..3 = this wavefront's four U instructions, 4..7 = its V instructions
  auto issue1 = [&](int t, int idx) {
    char* st = lds + (t % NST) * STAGE;
    const long m = m_lo + (long)t * TK;
    const bool live = t < ntiles;
    const bool full = m + TK <= a.M;
    const int i = idx & 3;
    const bool ok = live & (full | (m + rw[i] < a.M));
    if (idx < 4) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsu, (__attribute__((address_space(3))) void*)(st + (4 * i + wave) * 1024), 16,
                                               ok ? vu[i] : TNTR_OOB, live ? (uint32_t)(m * a.ldu * 2) : 0u, 0, 0);
    } else {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsv, (__attribute__((address_space(3))) void*)(st + TILE_B + (4 * i + wave) * 1024), 16,
                                               ok ? vv[i] : TNTR_OOB, live ? (uint32_t)(m * a.ldv * 2) : 0u, 0, 0);
    }
  };
#pragma unroll
  for (int t = 0; t < NST - 1; ++t)
#pragma unroll
    for (int idx = 0; idx < TNTR_NI; ++idx) issue1(t, idx);
  for (int kt = 0; kt < ntiles; ++kt) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * TNTR_NI) : "memory");   // this wavefront's pieces of tile kt have landed
    __builtin_amdgcn_s_barrier();                                                // everyone's have; nobody reads tile kt-1 any more
    asm volatile("" ::: "memory");
    const char* sU = lds + (kt % NST) * STAGE;
    const char* sV = sU + TILE_B;
#ifndef TNTR_NO_MFMA
    // all 16 fragments of the tile first, then 8 groups of 4 MFMAs with ONE DMA piece of tile kt+NST-1 after each group: a
    // wavefront that issues its 8 pieces in a burst sits in the (saturated) DMA queue while its SIMD has nothing to run
    bf16x8_t fa[2][FM], fb[2][FN];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
      for (int i = 0; i < FM; ++i) fa[s2][i] = tr_frag(sU, s2, wm0 + i * 16, lane);
#pragma unroll
      for (int j = 0; j < FN; ++j) fb[s2][j] = tr_frag(sV, s2, wn0 + j * 16, lane);
    }
    constexpr int PER = TNTR_NI / (2 * FM);   // pieces per MFMA group: 1 (64x64 per wavefront) or 2 (32x32)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int i = 0; i < FM; ++i) {
#pragma unroll
        for (int j = 0; j < FN; ++j)
          acc[i][j] = TRANS ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[s2][j], fa[s2][i], acc[i][j], 0, 0, 0)
                            : __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[s2][i], fb[s2][j], acc[i][j], 0, 0, 0);
#ifndef TNTR_NO_DMA
#pragma unroll
        for (int q = 0; q < PER; ++q) issue1(kt + NST - 1, (s2 * FM + i) * PER + q);
#endif
        __builtin_amdgcn_sched_barrier(0);
      }
#else
#pragma unroll
    for (int idx = 0; idx < TNTR_NI; ++idx) issue1(kt + NST - 1, idx);
    if (sV[lane] == 77 && kt == 1 << 30) acc[0][0][0] += 1.f;
#endif
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // trailing zero-fill DMAs still write LDS
  tn_tr_store<BQ, TRANS>(a, acc, p0, q0, wm0, wn0, lane);
}

// NST = 0: the register-prefetch body (32 KB of LDS); NST >= 2: the LDS-DMA ring (NST x 32 KB), unless an operand spans >= 1 GiB
template <int NST>
__device__ __forceinline__ void tn_tr_dispatch(const TnArgs& a, int tile, int split, char* lds) {
  bool dma = NST >= 2;
  if (NST >= 2) dma = (a.M * a.ldu * 2 < (long)TNTR_BUF) & (a.M * a.ldv * 2 < (long)TNTR_BUF);
  if constexpr (NST >= 2) {
    if (dma) {
      if (a.narrow) {
        if (a.trans) tn_tr_body_dma<32, true, NST>(a, tile, split, lds);
        else tn_tr_body_dma<32, false, NST>(a, tile, split, lds);
      } else {
        if (a.trans) tn_tr_body_dma<128, true, NST>(a, tile, split, lds);
        else tn_tr_body_dma<128, false, NST>(a, tile, split, lds);
      }
      return;
    }
  }
  char* sU = lds;
  char* sV = lds + TK * PITCH;
  if (a.narrow) {
    if (a.trans) tn_tr_body<32, true>(a, tile, split, sU, sV);
    else tn_tr_body<32, false>(a, tile, split, sU, sV);
  } else {
    if (a.trans) tn_tr_body<128, true>(a, tile, split, sU, sV);
    else tn_tr_body<128, false>(a, tile, split, sU, sV);
  }
}

// Hardware block b runs on XCD b % 8.  The workgroups that share operand rows (the tiles of ONE token split: every P tile re-reads
// the split's V rows, every Q tile its U rows) must share an L2, or each XCD fetches its own copy over the fabric (measured: 923 MB
// through the fabric for 189 MB of operands on 32768 x 2560 x 320).  Logical blocks are numbered split-major / tile-minor and XCD x
// takes a contiguous range of them.
__device__ __forceinline__ int xcd_contiguous(int bid, int nblk) {
  const int qq = nblk >> 3, rr = nblk & 7, xcd = bid & 7;
  return (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
}

template <int NST, int OCC>
__global__ __launch_bounds__(256, OCC) void gemm_tn_tr_kernel(const TnArgs a, int tiles) {
  __shared__ __attribute__((aligned(16))) char lds[(NST >= 2 ? NST : 1) * 2 * TK * PITCH];
  const int l = a.narrow ? (int)blockIdx.x : xcd_contiguous(blockIdx.x, gridDim.x);   // 128x32 tiles share no operand rows
  tn_tr_dispatch<NST>(a, l % tiles, l / tiles, lds);
}

// Grouped form: ONE launch for all wide weight gradients of a backward pass (or of one exchange bucket).  The table is
// an array of TnTrDesc in device memory; first_block is a running prefix, block_base offsets a sub-range launch.
struct TnTrDesc {
  TnArgs a;         // 88 bytes
  int first_block;  // workgroups of all earlier descriptors
  int n_tiles;      // tiles_p * tiles_q
};
static_assert(offsetof(TnTrDesc, first_block) == 88, "ops.DeferredDW.KINDS patches first_block at byte 88");
static_assert(sizeof(TnArgs) == 88 && sizeof(TnTrDesc) == 96, "descriptor layout is part of the ABI");

template <int NST, int OCC>
__global__ __launch_bounds__(256, OCC) void gemm_tn_tr_grouped_kernel(const TnTrDesc* __restrict__ descs, int n, int block_base,
                                                                      int remap) {
  __shared__ __attribute__((aligned(16))) char lds[(NST >= 2 ? NST : 1) * 2 * TK * PITCH];
  // the mapping must be one bijection for the whole launch: the first problem decides (a table holds the gradients of ONE rank:
  // all 128x32 tiles -- no shared operand rows, the round-robin order balances better: +0.1 ms with the contiguous one at rank
  // 32 -- or all 128x128, -0.5 ms at rank 320)
  const bool contiguous = remap < 0 ? !descs[0].a.narrow : remap != 0;   // remap: -1 = by the first problem, 0 / 1 = A/B hook
  const int bid = (contiguous ? xcd_contiguous(blockIdx.x, gridDim.x) : (int)blockIdx.x) + block_base;
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (descs[mid].first_block <= bid) lo = mid; else hi = mid - 1;
  }
  const TnTrDesc d = descs[lo];
  const int local = bid - d.first_block;
  // split-major / tile-minor: the tiles of one token split are consecutive logical blocks, i.e. on one XCD (see xcd_contiguous)
  tn_tr_dispatch<NST>(d.a, local % d.n_tiles, local / d.n_tiles, lds);
}

// Round 6: the 128 x 160 tile for problems with a side of 320 (the LoRA rank of BASELINE config 3: dB [C, 320] = dY^T.Ts and
// dA [320, K] = dT^T.X; with 128-wide tiles 320 pads to 384 and 17 % of the MFMAs multiply zeros, and the wider operand is fetched once
// per THREE column tiles instead of two).  The 320 side is always the Q side (a 320 on the P side swaps the operands and writes the
// result transposed, like the rank <= 32 tiles): 2 x 2 wavefronts of 64 x 80, 20 accumulator fragments, V tile rows of 512 bytes.
// Its own kernel (two workgroups per CU: 80 accumulator registers + 5 staging slots do not fit the 168 of three per CU) and its own
// descriptor table (ops.DeferredDW kind "x").
template <int OCC>
__global__ __launch_bounds__(256, OCC) void gemm_tn_tr160_grouped_kernel(const TnTrDesc* __restrict__ descs, int n, int block_base) {
  __shared__ __attribute__((aligned(16))) char lds[TK * PITCH + TK * PITCH160];
  const int bid = xcd_contiguous(blockIdx.x, gridDim.x) + block_base;
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (descs[mid].first_block <= bid) lo = mid; else hi = mid - 1;
  }
  const TnTrDesc d = descs[lo];
  const int local = bid - d.first_block;
  if (d.a.trans) tn_tr_body<160, true>(d.a, local % d.n_tiles, local / d.n_tiles, lds, lds + TK * PITCH);
  else tn_tr_body<160, false>(d.a, local % d.n_tiles, local / d.n_tiles, lds, lds + TK * PITCH);
}

// body of the launch: AQL_TNTR_NST = 0 (register prefetch, 2 workgroups per CU; the default), 2 (DMA ring of 64 KB, 2 per CU), 3 / 4
// (96 / 128 KB, 1 per CU).  Measured (profiles/r02_tntr_dma_ring.txt): alone, on rotating operands, the 2-stage ring is 10-16 %
// faster than the register body (32768 x 2560 x 320: 141 -> 111 us) and one workgroup per CU with a deeper ring much slower
// (189 us: the loop is bound by the ~40 B/clk a CU takes in, not by bytes in flight); inside the rank-320 train step the ring is
// 1.0 ms SLOWER than the register body (52.1 vs 51.2 ms on one box), so the register body stays.
// workgroups per CU of the register body: 3 (168 VGPRs, no scratch; the default: config 3 49.53 -> 49.11 ms on one box, config 2
// unchanged) or 2 (196 VGPRs); 4 would spill 213 VGPRs
inline int tn_tr_occ() {
  static const int v = AQL_TUNE_INT("AQL_TNTR_OCC", 3);
  return v;
}
inline int tn_tr_nst() {
  static const int v = getenv("AQL_TNTR_NST") ? atoi(getenv("AQL_TNTR_NST")) : 0;
  return v;
}

// token tiles per workgroup: enough to amortise the prologue and the 128x128 fp32 atomic epilogue
inline int tn_tr_splits(int tiles, int ktiles, bool grouped) {
  static const int force = getenv("AQL_TN_SPLITS") ? atoi(getenv("AQL_TN_SPLITS")) : 0;
  int splits;
  if (force > 0) splits = force;
  else if (grouped) {
    static const int tps = AQL_TUNE_INT("AQL_TN_TPS", 32);   // token tiles per workgroup (tuning hook)
    splits = ktiles / tps;                              // the launch as a whole fills the chip
  }
  else splits = (320 + tiles - 1) / tiles;           // a lone problem: about one workgroup per CU (measured optimum)
  if (splits > ktiles / 2) splits = ktiles / 2;
  if (splits < 1) splits = 1;
  return splits;
}

inline bool tn_tr_fill(TnArgs* a, const bf16_t* U, long ldu, const bf16_t* V, long ldv, long M, int P, int Q, float alpha,
                       float* C, long ldc, bool grouped, int* n_tiles, int* n_blocks) {
  if (!U || !V || !C || M <= 0 || P <= 0 || Q <= 0 || P % 8 || Q % 8 || ldu % 8 || ldv % 8) return false;
  if ((((uintptr_t)U | (uintptr_t)V) & 15) != 0) return false;
  // C[P,Q] = U^T V.  A rank <= 32 side becomes the 32-wide Q side of 128x32 tiles; if that side is P the operands are
  // swapped (C^T = V^T U) and the result is written transposed.
  const bool swap = P <= 32 && Q > 32;
  a->narrow = (P <= 32 || Q <= 32) ? 1 : 0;
  a->trans = swap ? 1 : 0;
  a->U = swap ? V : U; a->V = swap ? U : V;
  a->ldu = swap ? ldv : ldu; a->ldv = swap ? ldu : ldv;
  a->M = M; a->P = swap ? Q : P; a->Q = swap ? P : Q;
  a->alpha = alpha; a->C = C; a->ldc = ldc;
  const int bq = a->narrow ? 32 : BT;
  a->tiles_q = aql_cdiv(a->Q, bq);
  const int tiles = aql_cdiv(a->P, BT) * a->tiles_q;
  const int ktiles = aql_cdiv(M, TK);
  int splits = tn_tr_splits(tiles, ktiles, grouped);
  a->tiles_per_split = aql_cdiv(ktiles, splits);
  splits = aql_cdiv(ktiles, a->tiles_per_split);
  *n_tiles = tiles;
  *n_blocks = tiles * splits;
  return true;
}

// the 128 x 160 form takes C[P,Q] = U^T V when exactly one side is a multiple of 160 that 128 does not divide (320, 960)
inline bool tn_tr_fill160(TnArgs* a, const bf16_t* U, long ldu, const bf16_t* V, long ldv, long M, int P, int Q, float alpha,
                          float* C, long ldc, int* n_tiles, int* n_blocks) {
  if (!U || !V || !C || M <= 0 || P <= 32 || Q <= 32 || P % 8 || Q % 8 || ldu % 8 || ldv % 8) return false;
  if ((((uintptr_t)U | (uintptr_t)V) & 15) != 0) return false;
  const bool q160 = Q % 160 == 0 && Q % 128 != 0, p160 = P % 160 == 0 && P % 128 != 0;
  if (!q160 && !p160) return false;
  const bool swap = !q160;
  a->narrow = 2;
  a->trans = swap ? 1 : 0;
  a->U = swap ? V : U; a->V = swap ? U : V;
  a->ldu = swap ? ldv : ldu; a->ldv = swap ? ldu : ldv;
  a->M = M; a->P = swap ? Q : P; a->Q = swap ? P : Q;
  a->alpha = alpha; a->C = C; a->ldc = ldc;
  a->tiles_q = a->Q / 160;
  const int tiles = aql_cdiv(a->P, BT) * a->tiles_q;
  const int ktiles = aql_cdiv(M, TK);
  int splits = tn_tr_splits(tiles, ktiles, true);
  a->tiles_per_split = aql_cdiv(ktiles, splits);
  splits = aql_cdiv(ktiles, a->tiles_per_split);
  *n_tiles = tiles;
  *n_blocks = tiles * splits;
  return true;
}

}  // namespace

// The 128 x 160 tile (round 6): descriptor fill (0 = the problem has no side of 320 / 960: the caller tries aql_tntr_desc_fill) and
// the grouped launch over a table of such descriptors.
extern "C" int aql_tntr160_desc_fill(void* host_desc, const bf16_t* U, long ldu, const bf16_t* V, long ldv, long M, int P, int Q,
                                     float alpha, float* C, long ldc, int first_block) {
  static const int en = getenv("AQL_TNTR160") ? atoi(getenv("AQL_TNTR160")) : 1;   // A/B hook: 0 = 128 x 128 tiles everywhere
  if (host_desc == nullptr || !en) return 0;
  TnTrDesc d;
  memset(&d, 0, sizeof(d));
  int n_tiles = 0, n_blocks = 0;
  if (!tn_tr_fill160(&d.a, U, ldu, V, ldv, M, P, Q, alpha, C, ldc, &n_tiles, &n_blocks)) return 0;
  d.first_block = first_block;
  d.n_tiles = n_tiles;
  memcpy(host_desc, &d, sizeof(d));
  return n_blocks;
}

extern "C" int aql_gemm_tn_tr160_grouped(const void* dev_descs, int first, int n, int block_base, int n_blocks, hipStream_t stream) {
  AQL_CHECK_ARG(dev_descs && first >= 0 && n > 0 && block_base >= 0 && n_blocks > 0, "aql_gemm_tn_tr160_grouped: bad args");
  const TnTrDesc* dd = static_cast<const TnTrDesc*>(dev_descs) + first;
  hipLaunchKernelGGL((gemm_tn_tr160_grouped_kernel<2>), dim3(n_blocks), dim3(256), 0, stream, dd, n, block_base);
  AQL_CHECK_LAUNCH("aql_gemm_tn_tr160_grouped");
  return AQL_OK;
}

// host-side descriptor (96 bytes) for the grouped launch; returns the workgroups it needs, 0 if the problem is not
// eligible (caller launches it on its own)
extern "C" int aql_tntr_desc_fill(void* host_desc, const bf16_t* U, long ldu, const bf16_t* V, long ldv, long M, int P, int Q,
                                  float alpha, float* C, long ldc, int first_block) {
  if (host_desc == nullptr) return 0;
  TnTrDesc d;
  memset(&d, 0, sizeof(d));
  int n_tiles = 0, n_blocks = 0;
  if (!tn_tr_fill(&d.a, U, ldu, V, ldv, M, P, Q, alpha, C, ldc, true, &n_tiles, &n_blocks)) return 0;
  d.first_block = first_block;
  d.n_tiles = n_tiles;
  memcpy(host_desc, &d, sizeof(d));
  return n_blocks;
}

// descriptors [first, first+n) of a device table; block_base = first_block of descriptor `first`
extern "C" int aql_gemm_tn_tr_grouped(const void* dev_descs, int first, int n, int block_base, int n_blocks,
                                      hipStream_t stream) {
  AQL_CHECK_ARG(dev_descs && first >= 0 && n > 0 && block_base >= 0 && n_blocks > 0, "aql_gemm_tn_tr_grouped: bad args");
  const TnTrDesc* dd = static_cast<const TnTrDesc*>(dev_descs) + first;
  static const int remap = AQL_TUNE_INT("AQL_TNTR_REMAP", -1);
  switch (tn_tr_nst()) {
    case 0:
      if (tn_tr_occ() == 3) hipLaunchKernelGGL((gemm_tn_tr_grouped_kernel<0, 3>), dim3(n_blocks), dim3(256), 0, stream, dd, n, block_base, remap);
      else hipLaunchKernelGGL((gemm_tn_tr_grouped_kernel<0, 2>), dim3(n_blocks), dim3(256), 0, stream, dd, n, block_base, remap);
      break;
    case 2: hipLaunchKernelGGL((gemm_tn_tr_grouped_kernel<2, 2>), dim3(n_blocks), dim3(256), 0, stream, dd, n, block_base, remap); break;
    case 3: hipLaunchKernelGGL((gemm_tn_tr_grouped_kernel<3, 1>), dim3(n_blocks), dim3(256), 0, stream, dd, n, block_base, remap); break;
    default: hipLaunchKernelGGL((gemm_tn_tr_grouped_kernel<4, 1>), dim3(n_blocks), dim3(256), 0, stream, dd, n, block_base, remap); break;
  }
  AQL_CHECK_LAUNCH("aql_gemm_tn_tr_grouped");
  return AQL_OK;
}

extern "C" int aql_gemm_tn_tr_f32(const bf16_t* U, long ldu, const bf16_t* V, long ldv, long M, int P, int Q, float alpha,
                                  float* C, long ldc, hipStream_t stream) {
  AQL_CHECK_ARG(U && V && C, "aql_gemm_tn_tr_f32: null operand");
  AQL_CHECK_ARG(M > 0 && P > 0 && Q > 0 && P % 8 == 0 && Q % 8 == 0 && ldu % 8 == 0 && ldv % 8 == 0,
                "aql_gemm_tn_tr_f32: P, Q and leading dimensions must be multiples of 8 (P=%d Q=%d)", P, Q);
  AQL_CHECK_ARG((((uintptr_t)U | (uintptr_t)V) & 15) == 0, "aql_gemm_tn_tr_f32: operands must be 16-byte aligned");
  TnArgs a;
  int tiles = 0, n_blocks = 0;
  AQL_CHECK_ARG(tn_tr_fill(&a, U, ldu, V, ldv, M, P, Q, alpha, C, ldc, false, &tiles, &n_blocks), "aql_gemm_tn_tr_f32: bad problem");
  const int splits = n_blocks / tiles;
  switch (tn_tr_nst()) {
    case 0:
      if (tn_tr_occ() == 3) hipLaunchKernelGGL((gemm_tn_tr_kernel<0, 3>), dim3(tiles * splits), dim3(256), 0, stream, a, tiles);
      else hipLaunchKernelGGL((gemm_tn_tr_kernel<0, 2>), dim3(tiles * splits), dim3(256), 0, stream, a, tiles);
      break;
    case 2: hipLaunchKernelGGL((gemm_tn_tr_kernel<2, 2>), dim3(tiles * splits), dim3(256), 0, stream, a, tiles); break;
    case 3: hipLaunchKernelGGL((gemm_tn_tr_kernel<3, 1>), dim3(tiles * splits), dim3(256), 0, stream, a, tiles); break;
    default: hipLaunchKernelGGL((gemm_tn_tr_kernel<4, 1>), dim3(tiles * splits), dim3(256), 0, stream, a, tiles); break;
  }
  AQL_CHECK_LAUNCH("aql_gemm_tn_tr_f32");
  return AQL_OK;
}
