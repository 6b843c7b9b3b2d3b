// bf16 MFMA GEMM core for gfx950 (CDNA4).  Operand descriptors ("loaders") are shared by three kernel bodies:
//   * gemm_body_w  -- wave-specialised LDS-DMA kernel (4 loader + 4 MFMA wavefronts, 16x16x32 MFMA, 160-wide tiles):
//                     the fused LoRA linear  Y = X.W^T + b + (T*S).Bup^T  (two K segments, one accumulator), the 3x3
//                     NHWC convolution as an implicit GEMM (gather loader, nearest-x2 upsample and stride folded in) and
//                     its backward-data form, whenever the grid is a whole number of one-workgroup-per-CU rounds;
//   * gemm_body_d  -- the same LDS-DMA pipeline with 4 all-purpose wavefronts for short-K / odd-grid problems;
//   * gemm_body    -- 32x32x16-MFMA kernel with a register-transposing loader for the token-reduction GEMMs dA / dB
//                     (rank <= 32 LoRA weight gradients, grouped launch, fp32 atomics).
// Common to all: the *weight-side* tile is MFMA operand A and the *activation-side* tile operand B, so a lane ends up
// with 4 consecutive output columns of one output row; LDS rows are 128 B (8 chunks of 16 B) with the chunk index
// XOR-swizzled by (row >> 1) & 7 so that staging and ds_read_b128 fragment reads are bank-conflict free.
#pragma once
#include <type_traits>
#include "aql_common.h"

namespace aqlgemm {

constexpr int BK = 64;
constexpr int NTHREADS = 256;

// ds_read_b128 is serviced in four NON-contiguous 16-lane groups ({0-3,12-15,20-27}, {4-11,16-19,28-31}, +32) over a
// 256-B bank window (two 128-B rows).  XOR-ing the chunk with (row>>1)&7 gives every group 8 distinct slots per row
// parity, i.e. conflict-free fragment reads; the (row&7) swizzle measured 33 % SQ_LDS_BANK_CONFLICT (rows r and r+8
// of one group collide).
__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

__device__ __forceinline__ uint4 zero4() { return make_uint4(0u, 0u, 0u, 0u); }

// ------------------------------------------------------------------------------------------------
// Loaders.  A loader describes one GEMM operand as "rows x K" and hands out 16-byte chunks (8 bf16
// along K).  kTrans loaders read a [K][rows] source instead and are staged through a register transpose.
// ------------------------------------------------------------------------------------------------
struct PlainLoader {
  static constexpr bool kTrans = false;
  const bf16_t* base;
  long ld;
  int rows;
  int K;
  // GEGLU tiles (EpiParams::geglu_F): tile-local row lr >= gsplit reads source row (row0 + lr + goff), so that one 160-row
  // weight tile holds 80 "value" rows [n0, n0+80) and the matching 80 "gate" rows [F+n0, F+n0+80).  0 = plain rows.
  int gsplit = 0, goff = 0;
  int row_lo = 0;  // rows below row_lo read as zeros (segment 1 of a twin batch: the clean half has no LoRA term)
  // Per-sample operand (the weight side of a GEMM whose weights differ per sample: W + Bup.diag(S_b).A, aql_gemm_bf16_sw): the
  // workgroup whose first output row is m0 reads sample s = m0 / srows - s0 of `sbase` (stride `sstride` elements) instead of
  // `base` when s >= 0.  srows is a multiple of every tile height, so a tile never straddles two samples.  srows = 0: off.
  const bf16_t* sbase = nullptr;
  long sstride = 0;
  int srows = 0, s0 = 0;
};

// the weight-side loader of the workgroup whose output tile starts at row m0 (see PlainLoader::sbase)
template <class LB>
__device__ __forceinline__ LB sample_operand(const LB& l, int m0) {
  LB r = l;
  if constexpr (std::is_same<LB, PlainLoader>::value) {
    // block-uniform (m0 comes from the block index): kept in scalar registers through readfirstlane, the buffer descriptor built
    // from the result must be wave-uniform
    const int rows = l.srows > 0 ? l.srows : 1;
    const int s = __builtin_amdgcn_readfirstlane(m0 / rows - l.s0);
    const bool use = l.srows > 0 && s >= 0;
    const long off = use ? (long)s * l.sstride : 0;
    r.base = (use ? l.sbase : l.base) + off;
  }
  return r;
}

// Forward 3x3 conv, NHWC input [B,Hin,Win,Cin]; row r = (b,ho,wo); k = (kh*3+kw)*Cin + ci.
// ups=1 reads the input through a nearest x2 upsample (diffusers Upsample2D) without materialising it.
struct ConvFwdLoader {
  static constexpr bool kTrans = false;
  const bf16_t* base;
  int B, Hin, Win, Cin, Hout, Wout, stride, ups;
  int rows, K;
  int pad;  // leading (top/left) zero padding: 1, or 0 for the VAE's stride-2 convs that pad (0,1,0,1)
};

// Backward-data 3x3 conv: rows are input pixels (b,hi,wi) of dX; source is dY [B,Hout,Wout,Cout];
// k = (kh*3+kw)*Cout + co;  hi = ho*stride + kh - 1.
struct ConvBwdLoader {
  static constexpr bool kTrans = false;
  const bf16_t* base;
  int B, Hin, Win, Cout, Hout, Wout, stride;
  int rows, K;
};

// Transposed source: element (row, k) lives at base[k*ld + row]  (rows contiguous).  Used for the
// token-reduction GEMMs dB = dY^T.Ts and dA = dT^T.X.  rows % 8 == 0 required.
struct TransLoader {
  static constexpr bool kTrans = true;
  const bf16_t* base;
  long ld;
  int rows;
  int K;
};

// ------------------------------------------------------------------------------------------------
// Epilogue descriptors
// ------------------------------------------------------------------------------------------------
enum { EPI_BF16 = 0, EPI_SLAB = 1, EPI_ATOMIC = 2 };

struct EpiParams {
  // EPI_BF16: C[m][n] = bf16(bf16(acc + bias[n]) + residual[m][n]);  optional C2 = bf16(C * rowscale[m/rps][n])
  bf16_t* C;
  long ldc;
  const bf16_t* bias;      // [N] or null
  const bf16_t* residual;  // [M][ldr] or null
  long ldr;
  int res_mod = 0;         // > 0: row m reads residual row m % res_mod (the same [res_mod][N] block under every sample of a stacked M)
  bf16_t* C2;              // second output (row-scaled) or null
  long ldc2;
  const bf16_t* rowscale;  // [nsamples][N]
  const bf16_t* rowbias;   // [nsamples][rowbias_ld] added (bf16 add) after bias, before the residual; or null
  long rowbias_ld;
  int rows_per_sample;
  // GEGLU epilogue (scripts/lib/original_unet.py:727-729), geglu_F = F > 0: the GEMM's N = 2F output columns are produced as
  // [value | gate] column pairs inside one tile and  G[m][n] = value * gelu_erf(gate)  ([M][F]) is written; the pre-activation
  // C ([M][2F]) is written only when C != null (the backward pass needs it, the frozen pass does not).
  bf16_t* G = nullptr;
  long ldg = 0;
  int geglu_F = 0;
  int c_row0 = 0;          // GEGLU: the pre-activation C is written for rows >= c_row0 only (twin batches: the clean half
                           // never runs backward)
  // GEGLU-backward epilogue (gb_F = F > 0; the one-launch LoRA linear in its backward-data form, ff.net.2): the GEMM's N = F
  // outputs are d(activated) = d(value * gelu(gate)); with the saved pre-activation gb_h [M][2F] the epilogue writes
  // d(value) to C[m][n] and d(gate) to C[m][F + n] (ldc = 2F) -- aql_geglu_bwd applied to the bf16-rounded tile.
  const bf16_t* gb_h = nullptr;
  long gb_ldh = 0;
  int gb_F = 0;
  int trans_out;           // EPI_ATOMIC only: write C^T, i.e. element (m,n) goes to Cf[n*ldcf + m]
  // EPI_SLAB / EPI_ATOMIC: fp32 output
  float* Cf;               // slab base [splits][M][ldcf] or atomic target [M][ldcf]
  long ldcf;
  float alpha;             // scale applied to fp32 outputs
};

struct Segment {
  int ktiles;  // number of BK tiles of this K segment
};

template <class LA, class LB>
struct GemmArgs {
  LA a0;  // activation-side operand (rows = M), segment 0
  LB b0;  // weight-side operand (rows = N), segment 0
  LA a1;  // optional segment 1 (LoRA:  Ts [M,r]  x  Bup [N,r])
  LB b1;
  int ktiles0, ktiles1;
  int M, N;
  int seg1_row0 = 0;  // output tiles that end at or below this row skip K segment 1 (twin batch: clean rows carry no LoRA term)
  int splits;  // grid.z; k tiles of the concatenated K range are divided evenly
  int m_fast;  // tile order: 0 = N tiles of one M tile adjacent (activation reuse), 1 = M tiles adjacent (weight reuse)
  // Column groups (round 6; 0 = off): output columns [g grp_n, (g + 1) grp_n) read the activation-side operand of segment 0 / 1 at a
  // column offset of g grp_a0 / g grp_a1 elements -- block-diagonal products as ONE GEMM: q | k | v of a rank-r LoRA read their own
  // slice of the stacked Ts [M, 3r] (segment 1), the three backward "down" products their own slice of [dQ | dK | dV] (segment 0).
  // grp_n is a multiple of every tile width the picker may choose for N (entry points check: multiples of 320).
  int grp_n = 0, grp_a0 = 0, grp_a1 = 0;
  EpiParams epi;
};

// the activation-side loader of the workgroup whose output tile starts at column n0 (GemmArgs::grp_n)
template <class LA>
__device__ __forceinline__ LA group_operand(const LA& l, int n0, int grp_n, int off) {
  LA r = l;
  if constexpr (std::is_same<LA, PlainLoader>::value) {
    const int gidx = __builtin_amdgcn_readfirstlane(grp_n > 0 ? n0 / grp_n : 0);   // block-uniform
    r.base = l.base + (long)gidx * off;
  }
  return r;
}

// --------------------------------------------------------------------------------------------
// staging helpers
// --------------------------------------------------------------------------------------------
template <int R, class L>
struct Stager;  // only the transposing specialisation below is used (token-reduction GEMMs)

template <int R>
struct Stager<R, TransLoader> {
  // transposed: task t -> kq = t&15 (4 consecutive k), rg = t>>4 (8 consecutive rows)
  static constexpr int TASKS = (R / 8) * 16;
  static constexpr int NL = (TASKS + NTHREADS - 1) / NTHREADS;
  uint4 regs[NL][4];
  int row0_;
  __device__ __forceinline__ void init(const TransLoader&, int row0, int) { row0_ = row0; }
  __device__ __forceinline__ void fetch(const TransLoader& l, int k0, int tid) {
#pragma unroll
    for (int t = 0; t < NL; ++t) {
      int task = tid + t * NTHREADS;
      int kq = task & 15, rg = task >> 4;
      int r = row0_ + rg * 8;
      bool rok = (task < TASKS) && (r < l.rows);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int k = k0 + kq * 4 + i;
        regs[t][i] = (rok && k < l.K) ? *reinterpret_cast<const uint4*>(l.base + (long)k * l.ld + r) : zero4();
      }
    }
  }
  __device__ __forceinline__ void commit(char* lds, int tid) const {
#pragma unroll
    for (int t = 0; t < NL; ++t) {
      int task = tid + t * NTHREADS;
      if (task >= TASKS) continue;
      int kq = task & 15, rg = task >> 4;
      const uint32_t* w0 = reinterpret_cast<const uint32_t*>(&regs[t][0]);
      const uint32_t* w1 = reinterpret_cast<const uint32_t*>(&regs[t][1]);
      const uint32_t* w2 = reinterpret_cast<const uint32_t*>(&regs[t][2]);
      const uint32_t* w3 = reinterpret_cast<const uint32_t*>(&regs[t][3]);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        int w = j >> 1;
        uint32_t lo, hi;
        if (j & 1) {
          lo = (w0[w] >> 16) | (w1[w] & 0xffff0000u);
          hi = (w2[w] >> 16) | (w3[w] & 0xffff0000u);
        } else {
          lo = (w0[w] & 0xffffu) | (w1[w] << 16);
          hi = (w2[w] & 0xffffu) | (w3[w] << 16);
        }
        int row = rg * 8 + j;
        *reinterpret_cast<uint2*>(lds + lds_off(row, kq >> 1) + (kq & 1) * 8) = make_uint2(lo, hi);
      }
    }
  }
};

// --------------------------------------------------------------------------------------------
// the kernel
// --------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, class LA, class LB, int EPI, int NSTG = 2>
__device__ __forceinline__ void gemm_body(const GemmArgs<LA, LB>& g, const int block_x, const int block_z) {
  constexpr int FM = WM / 32, FN = WN / 32;
  constexpr int WAVES_N = BN / WN;
  static_assert((BM / WM) * (BN / WN) == 4, "4 wavefronts per workgroup");
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
  constexpr int STAGE = A_BYTES + B_BYTES;
  constexpr int C_PITCH = (BN + 8) * 2;  // bytes per row of the bf16 C tile staged in LDS
  // NSTG = 2: double-buffered stages (one barrier per K tile).  NSTG = 1: a single stage and two barriers per K tile,
  // half the LDS -> twice the resident workgroups per CU, which is what the short-K, latency-bound shapes want.
  constexpr int LDS_BYTES = (NSTG * STAGE > BM * C_PITCH) ? NSTG * STAGE : BM * C_PITCH;
  __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm0 = (wave / WAVES_N) * WM;
  const int wn0 = (wave % WAVES_N) * WN;

  const int tiles_n = (g.N + BN - 1) / BN;
  const int tiles_m = (g.M + BM - 1) / BM;
  int tile_m, tile_n;
  if (g.m_fast) {
    tile_n = block_x / tiles_m;
    tile_m = block_x - tile_n * tiles_m;
  } else {
    tile_m = block_x / tiles_n;
    tile_n = block_x - tile_m * tiles_n;
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // K range of this split
  const int kt_total = g.ktiles0 + g.ktiles1;
  const int kt_begin = (int)(((long)kt_total * block_z) / g.splits);
  const int kt_end = (int)(((long)kt_total * (block_z + 1)) / g.splits);

  Stager<BM, LA> sa;
  Stager<BN, LB> sb;
  sa.init(g.a0, m0, tid);
  sb.init(g.b0, n0, tid);

  f32x16_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  auto fetch = [&](int kt) {
    if (kt < g.ktiles0) {
      sa.fetch(g.a0, kt * BK, tid);
      sb.fetch(g.b0, kt * BK, tid);
    } else {
      if (kt == g.ktiles0 || kt == kt_begin) {  // switch row descriptors to segment 1
        sa.init(g.a1, m0, tid);
        sb.init(g.b1, n0, tid);
      }
      sa.fetch(g.a1, (kt - g.ktiles0) * BK, tid);
      sb.fetch(g.b1, (kt - g.ktiles0) * BK, tid);
    }
  };

  if (kt_begin < kt_end) {
    fetch(kt_begin);
    sa.commit(lds, tid);
    sb.commit(lds + A_BYTES, tid);
  }
  __syncthreads();

  for (int kt = kt_begin; kt < kt_end; ++kt) {
    const int cur = (NSTG == 2) ? ((kt - kt_begin) & 1) : 0;
    char* sA = lds + cur * STAGE;
    char* sB = sA + A_BYTES;
    const bool more = (kt + 1 < kt_end);
    if (more) fetch(kt + 1);

#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      bf16x8_t fa[FM], fb[FN];
      const int chunk = ks * 2 + (lane >> 5);
#pragma unroll
      for (int i = 0; i < FM; ++i)
        fa[i] = *reinterpret_cast<const bf16x8_t*>(sA + lds_off(wm0 + i * 32 + (lane & 31), chunk));
#pragma unroll
      for (int j = 0; j < FN; ++j)
        fb[j] = *reinterpret_cast<const bf16x8_t*>(sB + lds_off(wn0 + j * 32 + (lane & 31), chunk));
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
    }

    if (NSTG == 1) __syncthreads();  // everybody is done reading the only stage
    if (more) {
      char* nA = lds + ((NSTG == 2) ? (cur ^ 1) : 0) * STAGE;
      sa.commit(nA, tid);
      sb.commit(nA + A_BYTES, tid);
    }
    __syncthreads();
  }

  // ---------------------------------------------------------------- epilogue
  // acc[i][j][e]: output row m = m0+wm0+i*32+(lane&31); col n = n0+wn0+j*32 + (e&3) + 8*(e>>2) + 4*(lane>>5)
  const EpiParams& ep = g.epi;
  if constexpr (EPI == EPI_BF16) {
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int row = wm0 + i * 32 + (lane & 31);
#pragma unroll
      for (int j = 0; j < FN; ++j) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int col = wn0 + j * 32 + q * 8 + (lane >> 5) * 4;
          float v0 = acc[i][j][q * 4 + 0], v1 = acc[i][j][q * 4 + 1];
          float v2 = acc[i][j][q * 4 + 2], v3 = acc[i][j][q * 4 + 3];
          if (ep.bias != nullptr && (n0 + col) < g.N) {
            const uint2 bb = *reinterpret_cast<const uint2*>(ep.bias + n0 + col);
            v0 += bf16lo(bb.x);
            v1 += bf16hi(bb.x);
            v2 += bf16lo(bb.y);
            v3 += bf16hi(bb.y);
          }
          *reinterpret_cast<uint2*>(lds + row * C_PITCH + col * 2) =
              make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
        }
      }
    }
    __syncthreads();
    constexpr int CPR = BN / 8;  // 16-byte chunks per row
    for (int id = tid; id < BM * CPR; id += NTHREADS) {
      const int row = id / CPR, cc = id - row * CPR;
      const int m = m0 + row, n = n0 + cc * 8;
      if (m >= g.M || n >= g.N) continue;
      uint4 v = *reinterpret_cast<const uint4*>(lds + row * C_PITCH + cc * 16);
      if (ep.rowbias != nullptr) {
        const uint4 r = *reinterpret_cast<const uint4*>(ep.rowbias + (long)(m / ep.rows_per_sample) * ep.rowbias_ld + n);
        v.x = pack_bf16x2(bf16lo(v.x) + bf16lo(r.x), bf16hi(v.x) + bf16hi(r.x));
        v.y = pack_bf16x2(bf16lo(v.y) + bf16lo(r.y), bf16hi(v.y) + bf16hi(r.y));
        v.z = pack_bf16x2(bf16lo(v.z) + bf16lo(r.z), bf16hi(v.z) + bf16hi(r.z));
        v.w = pack_bf16x2(bf16lo(v.w) + bf16lo(r.w), bf16hi(v.w) + bf16hi(r.w));
      }
      if (ep.residual != nullptr) {
        const uint4 r = *reinterpret_cast<const uint4*>(ep.residual + (long)m * ep.ldr + n);
        v.x = pack_bf16x2(bf16lo(v.x) + bf16lo(r.x), bf16hi(v.x) + bf16hi(r.x));
        v.y = pack_bf16x2(bf16lo(v.y) + bf16lo(r.y), bf16hi(v.y) + bf16hi(r.y));
        v.z = pack_bf16x2(bf16lo(v.z) + bf16lo(r.z), bf16hi(v.z) + bf16hi(r.z));
        v.w = pack_bf16x2(bf16lo(v.w) + bf16lo(r.w), bf16hi(v.w) + bf16hi(r.w));
      }
      if (ep.C != nullptr) *reinterpret_cast<uint4*>(ep.C + (long)m * ep.ldc + n) = v;
      if (ep.C2 != nullptr) {
        const uint4 s =
            *reinterpret_cast<const uint4*>(ep.rowscale + (long)(m / ep.rows_per_sample) * g.N + n);
        uint4 o;
        o.x = pack_bf16x2(bf16lo(v.x) * bf16lo(s.x), bf16hi(v.x) * bf16hi(s.x));
        o.y = pack_bf16x2(bf16lo(v.y) * bf16lo(s.y), bf16hi(v.y) * bf16hi(s.y));
        o.z = pack_bf16x2(bf16lo(v.z) * bf16lo(s.z), bf16hi(v.z) * bf16hi(s.z));
        o.w = pack_bf16x2(bf16lo(v.w) * bf16lo(s.w), bf16hi(v.w) * bf16hi(s.w));
        *reinterpret_cast<uint4*>(ep.C2 + (long)m * ep.ldc2 + n) = o;
      }
    }
  } else if constexpr (EPI == EPI_SLAB) {
    float* out = ep.Cf + (long)block_z * g.M * ep.ldcf;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int m = m0 + wm0 + i * 32 + (lane & 31);
#pragma unroll
      for (int j = 0; j < FN; ++j) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = n0 + wn0 + j * 32 + q * 8 + (lane >> 5) * 4;
          if (m >= g.M || n >= g.N) continue;
          *reinterpret_cast<float4*>(out + (long)m * ep.ldcf + n) =
              make_float4(acc[i][j][q * 4 + 0], acc[i][j][q * 4 + 1], acc[i][j][q * 4 + 2], acc[i][j][q * 4 + 3]);
        }
      }
    }
  } else {
    // EPI_ATOMIC: stage the fp32 tile in LDS so that consecutive lanes hit consecutive addresses of the target
    // (coalesced global_atomic_add_f32); trans_out writes C^T.
    static_assert(BM * BN * 4 <= LDS_BYTES, "fp32 tile must fit in the staging LDS");
    float* ct = reinterpret_cast<float*>(lds);
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int row = wm0 + i * 32 + (lane & 31);
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int col = wn0 + j * 32 + q * 8 + (lane >> 5) * 4;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            // [row][col] for the plain layout, [col][row] for the transposed one: the fast index is the output's
            if (ep.trans_out) ct[(col + e) * BM + row] = acc[i][j][q * 4 + e];
            else ct[row * BN + col + e] = acc[i][j][q * 4 + e];
          }
        }
    }
    __syncthreads();
    for (int id = tid; id < BM * BN; id += NTHREADS) {
      int row, col;
      if (ep.trans_out) {
        col = id / BM;
        row = id - col * BM;
      } else {
        row = id / BN;
        col = id - row * BN;
      }
      const int m = m0 + row, n = n0 + col;
      if (m >= g.M || n >= g.N) continue;
      const float v = ep.alpha * ct[id];
      if (ep.trans_out) atomicAdd(ep.Cf + (long)n * ep.ldcf + m, v);
      else atomicAdd(ep.Cf + (long)m * ep.ldcf + n, v);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// LDS-DMA main loop (non-transposed loaders).  Measured on MI355X (tools/tune_gemm.py, tools/trace_gemm.py):
//   * operands are read with raw BUFFER loads straight into LDS (buffer_load_dwordx4 ... lds): out-of-range rows / K
//     tails / conv padding become an out-of-range voffset that the hardware answers with zeros -- no branch, no staging
//     registers, no ds_write;
//   * 16x16x32 MFMA fragments, so BN = 160 tiles exist (every channel count of this U-Net is a multiple of 160);
//   * address generation is INCREMENTAL: an in-kernel trace showed the per-tile div/mod address math of the first
//     version costing 860 of 1950 cycles per K tile (a wave64 VALU op is 4 cycles, an integer division ~40 ops).  Tiles
//     are always requested in increasing order, so each stager keeps scalar (wave-uniform) tap / channel / byte-offset
//     state and spends <= 1 VALU op per weight row and ~6 per gathered activation row; the K offset of plain operands
//     rides in the instruction's scalar offset (not part of the hardware range check).
// ------------------------------------------------------------------------------------------------
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
constexpr uint32_t OOB_ROW = 0x80000000u;    // any voffset >= BUF_BYTES reads as zero
constexpr uint32_t BUF_BYTES = 0x40000000u;  // operands must be < 1 GiB (host-checked)
constexpr int FAR = 1 << 20;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, p ? BUF_BYTES : 0u, 0x00020000);
}
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, char* lds_dst, uint32_t voff, uint32_t soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_dst, 16, voff, soff, 0, 0);
}

// DmaStager<R, L>: the wave's 64 lanes fill 8 consecutive 128-B LDS rows per instruction; lane (row = 32i + tid>>3,
// slot = tid&7) fetches source chunk slot ^ ((row >> 1) & 7) (the XOR swizzle is applied on the SOURCE side), i.e.
// k element kc = ((tid&7) ^ ((tid>>4)&7)) * 8 of the tile.  begin() positions the stager on tile t_first of the
// concatenated K range; every dma() stages the next tile and advances.  Tiles at or past t_end are zero-filled.
// RPI = rows one piece instruction covers across the loader wavefronts: 8 rows per wavefront x 4 loaders = 32 (8 loaders: 64, experiment)
template <int R, class L, int RPI = 32>
struct DmaStager;

template <int R, int RPI>
struct DmaStager<R, PlainLoader, RPI> {
  static constexpr int NL = (R + RPI - 1) / RPI;
  uint32_t voff[NL], voff1[NL];  // row byte offset + lane chunk offset (current segment / segment 1), or OOB_ROW
  __amdgpu_buffer_rsrc_t rs, rs1;
  uint32_t soff;                 // byte offset of the next tile inside the current segment
  int krem, krem1;               // valid k elements from the next tile's first k on (<= 0: nothing valid)
  int to_switch;                 // tiles until segment 1 starts
  int kc;
  __device__ __forceinline__ void begin(const PlainLoader& l0, const PlainLoader& l1, bool dual, int row0, int tid,
                                        int t_first, int t_end, int ktiles0) {
    kc = ((tid & 7) ^ ((tid >> 4) & 7)) * 8;
    const bool in1 = dual && t_first >= ktiles0;
    rs1 = make_rsrc(dual ? l1.base : nullptr);
    rs = in1 ? rs1 : make_rsrc(l0.base);
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int lr = (tid >> 3) + RPI * i;
      const int r = row0 + lr + ((l0.gsplit && lr >= l0.gsplit) ? l0.goff : 0);
      const int r1 = row0 + lr + ((l1.gsplit && lr >= l1.gsplit) ? l1.goff : 0);
      const uint32_t v0 = r < l0.rows ? (uint32_t)r * (uint32_t)(l0.ld * 2) + kc * 2 : OOB_ROW;
      voff1[i] = (dual && r1 < l1.rows && r1 >= l1.row_lo) ? (uint32_t)r1 * (uint32_t)(l1.ld * 2) + kc * 2 : OOB_ROW;
      voff[i] = in1 ? voff1[i] : v0;
    }
    const int kend0 = min(l0.K, min(t_end, ktiles0) * BK);
    const int kend1 = dual ? min(l1.K, (t_end - ktiles0) * BK) : 0;
    const int tl = in1 ? t_first - ktiles0 : t_first;  // tile index inside the current segment
    soff = (uint32_t)tl * (BK * 2);
    krem = (in1 ? kend1 : kend0) - tl * BK;
    krem1 = kend1;
    to_switch = (dual && !in1) ? ktiles0 - t_first : 0x7fffffff;
  }
  __device__ __forceinline__ void dma(char* stage, int wave) {
    const bool bad = kc >= krem;  // K tail of the segment / past the end of this split
#pragma unroll
    for (int i = 0; i < NL; ++i)
      if (R % RPI == 0 || RPI * i + 8 * wave < R)   // wave-uniform: a piece past the tile's rows would zero-fill the next stage
        dma16(rs, stage + (RPI * i + 8 * wave) * 128, bad ? OOB_ROW : voff[i], soff);
    soff += BK * 2;
    krem -= BK;
    if (--to_switch == 0) {  // uniform: enter segment 1 (the LoRA rank segment)
      rs = rs1;
      soff = 0;
      krem = krem1;
#pragma unroll
      for (int i = 0; i < NL; ++i) voff[i] = voff1[i];
    }
  }
};

// 3x3 gather loaders.  Fast path (channels % 64 == 0, no folded upsample / stride-2 adjoint): a K tile lies inside one
// tap, so (kh, kw, c0) are scalars advanced per tile and a row's offset is rowbase + tapoff.  Anything else takes the
// general per-lane path (conv_in with 8 channels, the 3 upsample convs, the 3 stride-2 backward convs).
template <int R, int RPI>
struct DmaStager<R, ConvFwdLoader, RPI> {
  static constexpr int NL = (R + RPI - 1) / RPI;
  int rb[NL], rh[NL], rw[NL];  // b*Hin, h0, w0 of the row's output pixel (h0 = -FAR for rows past M)
  uint32_t rowbase[NL];        // fast path: (((b*Hin + h0)*Win + w0)*Cin + kc)*2   (wraps for border rows; only used when valid)
  __amdgpu_buffer_rsrc_t rs;
  int Cin, Win, Hl, Wl, ups, kc;
  int kh, kw, c0, left, t;     // scalar state of the next tile
  bool fast;
  __device__ __forceinline__ void begin(const ConvFwdLoader& l, const ConvFwdLoader&, bool, int row0, int tid,
                                        int t_first, int t_end, int ktiles0) {
    kc = ((tid & 7) ^ ((tid >> 4) & 7)) * 8;
    rs = make_rsrc(l.base);
    Cin = l.Cin, Win = l.Win, ups = l.ups, Hl = l.Hin << l.ups, Wl = l.Win << l.ups;
    fast = (l.Cin % BK == 0) && l.ups == 0;
    const int hw = l.Hout * l.Wout;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int r = row0 + (tid >> 3) + RPI * i;
      const int b = r / hw, rem = r - b * hw, ho = rem / l.Wout;
      rb[i] = b * l.Hin;
      rh[i] = r < l.rows ? ho * l.stride - l.pad : -FAR;
      rw[i] = (rem - ho * l.Wout) * l.stride - l.pad;
      rowbase[i] = (uint32_t)(((rb[i] + rh[i]) * Win + rw[i]) * Cin + kc) * 2u;
    }
    t = t_first;
    left = min(t_end, ktiles0) - t_first;  // tiles still inside the K range
    const int k0 = t_first * BK, tap = k0 / Cin;
    c0 = k0 - tap * Cin;
    kh = tap / 3;
    kw = tap - kh * 3;
  }
  __device__ __forceinline__ void dma(char* stage, int wave) {
    if (fast) {
      const int khe = left > 0 ? kh : -FAR;
      const uint32_t tapoff = (uint32_t)((khe * Win + kw) * Cin + c0) * 2u;
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        const bool ok = (unsigned)(rh[i] + khe) < (unsigned)Hl && (unsigned)(rw[i] + kw) < (unsigned)Wl;
        if (R % RPI == 0 || RPI * i + 8 * wave < R) dma16(rs, stage + (RPI * i + 8 * wave) * 128, ok ? rowbase[i] + tapoff : OOB_ROW, 0);
      }
      c0 += BK;
      if (c0 >= Cin) {
        c0 = 0;
        if (++kw == 3) kw = 0, ++kh;
      }
    } else {
      const int k = t * BK + kc;
      const bool kok = left > 0 && k < 9 * Cin;
      const int tap = k / Cin, ci = k - tap * Cin;
      const int khl = kok ? tap / 3 : -FAR, kwl = tap - (tap / 3) * 3;
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        const int hi = rh[i] + khl, wi = rw[i] + kwl;
        const bool ok = (unsigned)hi < (unsigned)Hl && (unsigned)wi < (unsigned)Wl;
        const uint32_t off = (uint32_t)(((rb[i] + (hi >> ups)) * Win + (wi >> ups)) * Cin + ci) * 2u;
        if (R % RPI == 0 || RPI * i + 8 * wave < R) dma16(rs, stage + (RPI * i + 8 * wave) * 128, ok ? off : OOB_ROW, 0);
      }
    }
    ++t;
    --left;
  }
};

template <int R, int RPI>
struct DmaStager<R, ConvBwdLoader, RPI> {
  static constexpr int NL = (R + RPI - 1) / RPI;
  int rb[NL], rh[NL], rw[NL];  // b*Hout, hi+1, wi+1 of the row's input pixel (hi+1 = -FAR for rows past M)
  uint32_t rowbase[NL];        // fast path: (((b*Hout + hi+1)*Wout + wi+1)*Cout + kc)*2
  __amdgpu_buffer_rsrc_t rs;
  int Cout, Hout, Wout, s2, kc;
  int kh, kw, c0, left, t;
  bool fast;
  __device__ __forceinline__ void begin(const ConvBwdLoader& l, const ConvBwdLoader&, bool, int row0, int tid,
                                        int t_first, int t_end, int ktiles0) {
    kc = ((tid & 7) ^ ((tid >> 4) & 7)) * 8;
    rs = make_rsrc(l.base);
    Cout = l.Cout, Hout = l.Hout, Wout = l.Wout, s2 = l.stride == 2 ? 1 : 0;
    fast = (l.Cout % BK == 0) && l.stride == 1;
    const int hw = l.Hin * l.Win;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int r = row0 + (tid >> 3) + RPI * i;
      const int b = r / hw, rem = r - b * hw, hi = rem / l.Win;
      rb[i] = b * l.Hout;
      rh[i] = r < l.rows ? hi + 1 : -FAR;
      rw[i] = (rem - hi * l.Win) + 1;
      rowbase[i] = (uint32_t)(((rb[i] + rh[i]) * Wout + rw[i]) * Cout + kc) * 2u;
    }
    t = t_first;
    left = min(t_end, ktiles0) - t_first;
    const int k0 = t_first * BK, tap = k0 / Cout;
    c0 = k0 - tap * Cout;
    kh = tap / 3;
    kw = tap - kh * 3;
  }
  __device__ __forceinline__ void dma(char* stage, int wave) {
    if (fast) {
      const int khe = left > 0 ? kh : 2 * FAR;
      const uint32_t tapoff = (uint32_t)(c0 - (khe * Wout + kw) * Cout) * 2u;
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        const bool ok = (unsigned)(rh[i] - khe) < (unsigned)Hout && (unsigned)(rw[i] - kw) < (unsigned)Wout;
        if (R % RPI == 0 || RPI * i + 8 * wave < R) dma16(rs, stage + (RPI * i + 8 * wave) * 128, ok ? rowbase[i] + tapoff : OOB_ROW, 0);
      }
      c0 += BK;
      if (c0 >= Cout) {
        c0 = 0;
        if (++kw == 3) kw = 0, ++kh;
      }
    } else {
      const int k = t * BK + kc;
      const bool kok = left > 0 && k < 9 * Cout;
      const int tap = k / Cout, co = k - tap * Cout;
      const int khl = kok ? tap / 3 : 2 * FAR, kwl = tap - (tap / 3) * 3;
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        const int th = rh[i] - khl, tw = rw[i] - kwl;
        const bool par = ((th | tw) & s2) == 0;  // stride 2: only even offsets hit an output pixel
        const int ho = th >> s2, wo = tw >> s2;
        const bool ok = par && th >= 0 && tw >= 0 && ho < Hout && wo < Wout;
        const uint32_t off = (uint32_t)(((rb[i] + ho) * Wout + wo) * Cout + co) * 2u;
        if (R % RPI == 0 || RPI * i + 8 * wave < R) dma16(rs, stage + (RPI * i + 8 * wave) * 128, ok ? off : OOB_ROW, 0);
      }
    }
    ++t;
    --left;
  }
};

// ---- GEGLU epilogue helpers ---------------------------------------------------------------------------------------
__device__ __forceinline__ float gelu_erf(float g) { return aql_gelu(g); }
// value * gelu(gate) on the two bf16 halves of a word, packed fp32 arithmetic (aql_gelu2)
__device__ __forceinline__ uint32_t geglu_word(uint32_t v, uint32_t gt) {
  const aql_f32x2_t a = aql_f32x2_t{bf16lo(v), bf16hi(v)} * aql_gelu2(aql_f32x2_t{bf16lo(gt), bf16hi(gt)});
  return pack_bf16x2(a.x, a.y);
}

// bias index of tile-local column `col` (a multiple of 4): plain tiles n0 + col; GEGLU tiles [value | gate] halves
__device__ __forceinline__ int epi_bias_col(int n0, int col, int F, int half) {
  return F == 0 ? n0 + col : (col < half ? n0 + col : F + n0 + col - half);
}

// aql_geglu_bwd on 8 elements: d = d(activated), hv / hg = saved value / gate -> dv = d gate cdf(gate), dg = d value (cdf + gate pdf)
__device__ __forceinline__ void geglu_bwd8(const uint4& d, const uint4& hv, const uint4& hg, uint4& dv, uint4& dg) {
  const uint32_t dw[4] = {d.x, d.y, d.z, d.w}, vw[4] = {hv.x, hv.y, hv.z, hv.w}, gw[4] = {hg.x, hg.y, hg.z, hg.w};
  uint32_t ov[4], og[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    aql_f32x2_t rv, rg;
    aql_geglu_bwd2(aql_f32x2_t{bf16lo(dw[e]), bf16hi(dw[e])}, aql_f32x2_t{bf16lo(vw[e]), bf16hi(vw[e])},
                   aql_f32x2_t{bf16lo(gw[e]), bf16hi(gw[e])}, rv, rg);
    ov[e] = pack_bf16x2(rv.x, rv.y);
    og[e] = pack_bf16x2(rg.x, rg.y);
  }
  dv = make_uint4(ov[0], ov[1], ov[2], ov[3]);
  dg = make_uint4(og[0], og[1], og[2], og[3]);
}

// The bf16 C tile [BM][BN] (value columns 0..BN/2-1, gate columns BN/2..BN-1) is staged in LDS: write
// G[m][n0 + c] = value * gelu(gate) and, when ep.C != null, the two pre-activation halves.  NT = participating threads.
template <int BM, int BN, int C_PITCH, int NT>
__device__ __forceinline__ void geglu_store(const char* lds, int m0, int n0, int M, const EpiParams& ep, int tid) {
  constexpr int HC = BN / 16;  // 16-byte chunks per half row
  const int F = ep.geglu_F;
  for (int id = tid; id < BM * HC; id += NT) {
    const int row = id / HC, cc = id - row * HC;
    const int m = m0 + row, n = n0 + cc * 8;
    if (m >= M || n >= F) continue;
    const uint4 v = *reinterpret_cast<const uint4*>(lds + row * C_PITCH + cc * 16);
    const uint4 gt = *reinterpret_cast<const uint4*>(lds + row * C_PITCH + BN + cc * 16);
    if (ep.C != nullptr && m >= ep.c_row0) {
      *reinterpret_cast<uint4*>(ep.C + (long)m * ep.ldc + n) = v;
      *reinterpret_cast<uint4*>(ep.C + (long)m * ep.ldc + F + n) = gt;
    }
    uint4 o;
    o.x = geglu_word(v.x, gt.x);
    o.y = geglu_word(v.y, gt.y);
    o.z = geglu_word(v.z, gt.z);
    o.w = geglu_word(v.w, gt.w);
    *reinterpret_cast<uint4*>(ep.G + (long)m * ep.ldg + n) = o;
  }
}

// ---- epilogue helpers shared by gemm_body_d / gemm_body_w / the one-launch LoRA kernels ----------------------------------
// A global load behind a per-lane branch cannot be hoisted past the branch, so the old epilogue -- `if (bias && col < N) load`
// inside the FM x FN fragment loop, `if (residual) load` inside the store loop -- ran every one of those loads as its own
// L2 round trip (load, s_waitcnt vmcnt(0), use): 20 + 10 dependent round trips per workgroup, ~10k + ~4k cycles of a 33k-cycle
// workgroup life on the short-K LoRA shapes (tools/trace_lora.py, MI355X).  Now the bias values of a lane's FN column groups
// are fetched BEFORE the K loop (unconditional loads: clamped address, AND-mask), and the store loop issues the residual /
// row-bias / row-scale / saved-activation loads of U items back to back before it touches any of them.
__device__ __forceinline__ uint2 epi_mask2(const uint2& v, bool ok) {
  const uint32_t m = 0u - (uint32_t)ok;
  return make_uint2(v.x & m, v.y & m);
}
__device__ __forceinline__ uint4 epi_mask4(const uint4& v, bool ok) {
  const uint32_t m = 0u - (uint32_t)ok;
  return make_uint4(v.x & m, v.y & m, v.z & m, v.w & m);
}
__device__ __forceinline__ uint4 epi_add8(const uint4& a, const uint4& b) {
  return make_uint4(pack_bf16x2(bf16lo(a.x) + bf16lo(b.x), bf16hi(a.x) + bf16hi(b.x)),
                    pack_bf16x2(bf16lo(a.y) + bf16lo(b.y), bf16hi(a.y) + bf16hi(b.y)),
                    pack_bf16x2(bf16lo(a.z) + bf16lo(b.z), bf16hi(a.z) + bf16hi(b.z)),
                    pack_bf16x2(bf16lo(a.w) + bf16lo(b.w), bf16hi(a.w) + bf16hi(b.w)));
}
__device__ __forceinline__ uint4 epi_mul8(const uint4& a, const uint4& b) {
  return make_uint4(pack_bf16x2(bf16lo(a.x) * bf16lo(b.x), bf16hi(a.x) * bf16hi(b.x)),
                    pack_bf16x2(bf16lo(a.y) * bf16lo(b.y), bf16hi(a.y) * bf16hi(b.y)),
                    pack_bf16x2(bf16lo(a.z) * bf16lo(b.z), bf16hi(a.z) * bf16hi(b.z)),
                    pack_bf16x2(bf16lo(a.w) * bf16lo(b.w), bf16hi(a.w) * bf16hi(b.w)));
}

// b[j] = bias[column group j of this lane] (4 bf16), zero when there is no bias or the group lies past N.  `safe` is any
// readable address (the weight panel): lanes without a value read it instead of branching.
template <int FN>
__device__ __forceinline__ void epi_load_bias(uint2 (&b)[FN], const bf16_t* bias, const void* safe, int n0, int wn0, int lane,
                                              int N, int gF, int half) {
  const bool has = bias != nullptr;
  const bf16_t* p = has ? bias : reinterpret_cast<const bf16_t*>(safe);
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int col = wn0 + j * 16 + (lane >> 4) * 4;
    const int bc = epi_bias_col(n0, col, gF, half);
    const bool ok = has & (bc < N);
    const uint2 v = *reinterpret_cast<const uint2*>(p + (ok ? bc : 0));
    b[j] = epi_mask2(v, ok);
  }
}

// The bf16 C tile [BM][BN] staged in LDS -> global, with the optional per-sample row bias, residual, second (row-scaled) output
// and the GEGLU-backward form (tile = d(value * gelu(gate)); writes d(value), d(gate) from the saved pre-activation).
template <int BM, int BN, int C_PITCH, int NT>
__device__ __forceinline__ void epi_store_tile(const char* lds, int m0, int n0, int M, int N, const EpiParams& ep, int tid) {
  constexpr int CPR = BN / 8, TOTAL = BM * CPR, NIT = (TOTAL + NT - 1) / NT;
  const bool has_rb = ep.rowbias != nullptr, has_res = ep.residual != nullptr, has_c2 = ep.C2 != nullptr, has_gb = ep.gb_F != 0;
  if (!(has_rb | has_res | has_c2 | has_gb)) {   // nothing to fetch: LDS -> global
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int id = tid + it * NT;
      const int row = id / CPR, cc = id - row * CPR;
      const int m = m0 + row, n = n0 + cc * 8;
      if (id < TOTAL && m < M && n < N && ep.C != nullptr)
        *reinterpret_cast<uint4*>(ep.C + (long)m * ep.ldc + n) = *reinterpret_cast<const uint4*>(lds + row * C_PITCH + cc * 16);
    }
    return;
  }
  // feature flags are workgroup-uniform: each batch of loads sits behind a scalar branch and is consumed behind the same one.
  // Two exclusive forms (GEGLU-backward / everything else) so that at most three 16-byte values per item are live: the batch
  // must fit beside the 4-wave kernels' AGPR accumulators at two workgroups per CU (<= 160 VGPRs).
  constexpr int U = NIT < 4 ? NIT : 4;   // items in flight per batch
#pragma unroll
  for (int it0 = 0; it0 < NIT; it0 += U) {
    bool ok[U];
    long om[U];
    int nn[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int id = tid + (it0 + u) * NT;
      const int row = id / CPR, cc = id - row * CPR;
      const int m = m0 + row, n = n0 + cc * 8;
      ok[u] = (it0 + u < NIT) & (id < TOTAL) & (m < M) & (n < N);
      om[u] = ok[u] ? m : 0;   // lanes without an item read element 0 of the operand instead of branching
      nn[u] = ok[u] ? n : 0;
    }
    if (has_gb) {
      uint4 rs[U], hv[U], hg[U];
      if (has_res) {
#pragma unroll
        for (int u = 0; u < U; ++u) rs[u] = *reinterpret_cast<const uint4*>(ep.residual + (ep.res_mod ? om[u] % ep.res_mod : om[u]) * ep.ldr + nn[u]);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        hv[u] = *reinterpret_cast<const uint4*>(ep.gb_h + om[u] * ep.gb_ldh + nn[u]);
        hg[u] = *reinterpret_cast<const uint4*>(ep.gb_h + om[u] * ep.gb_ldh + ep.gb_F + nn[u]);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int id = tid + (it0 + u) * NT;
        const int row = id / CPR, cc = id - row * CPR;
        if (!ok[u]) continue;
        uint4 v = *reinterpret_cast<const uint4*>(lds + row * C_PITCH + cc * 16);
        if (has_res) v = epi_add8(v, rs[u]);
        uint4 dv, dg;
        geglu_bwd8(v, hv[u], hg[u], dv, dg);
        *reinterpret_cast<uint4*>(ep.C + om[u] * ep.ldc + nn[u]) = dv;
        *reinterpret_cast<uint4*>(ep.C + om[u] * ep.ldc + ep.gb_F + nn[u]) = dg;
      }
    } else {
      uint4 rb[U], rs[U], sc[U];
      int smp[U];
      if (has_rb | has_c2) {
#pragma unroll
        for (int u = 0; u < U; ++u) smp[u] = (int)((uint32_t)om[u] / (uint32_t)ep.rows_per_sample);   // M < 2^31: 32-bit division
      }
      if (has_rb) {
#pragma unroll
        for (int u = 0; u < U; ++u) rb[u] = *reinterpret_cast<const uint4*>(ep.rowbias + (long)smp[u] * ep.rowbias_ld + nn[u]);
      }
      if (has_res) {
#pragma unroll
        for (int u = 0; u < U; ++u) rs[u] = *reinterpret_cast<const uint4*>(ep.residual + (ep.res_mod ? om[u] % ep.res_mod : om[u]) * ep.ldr + nn[u]);
      }
      if (has_c2) {
#pragma unroll
        for (int u = 0; u < U; ++u) sc[u] = *reinterpret_cast<const uint4*>(ep.rowscale + (long)smp[u] * N + nn[u]);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int id = tid + (it0 + u) * NT;
        const int row = id / CPR, cc = id - row * CPR;
        if (!ok[u]) continue;
        uint4 v = *reinterpret_cast<const uint4*>(lds + row * C_PITCH + cc * 16);
        if (has_rb) v = epi_add8(v, rb[u]);
        if (has_res) v = epi_add8(v, rs[u]);
        if (ep.C != nullptr) *reinterpret_cast<uint4*>(ep.C + om[u] * ep.ldc + nn[u]) = v;
        if (has_c2) *reinterpret_cast<uint4*>(ep.C2 + om[u] * ep.ldc2 + nn[u]) = epi_mul8(v, sc[u]);
      }
    }
  }
}

template <int BM, int BN, int WM, int WN, class LA, class LB, int EPI, int NSTG, int ABL = 0>
__device__ __forceinline__ void gemm_body_d(const GemmArgs<LA, LB>& g, const int block_x, const int block_z) {
  static_assert(EPI != EPI_ATOMIC && !LA::kTrans, "bf16 / slab epilogues only");
  constexpr int FM = WM / 16, FN = WN / 16;
  constexpr int WAVES_N = BN / WN;
  static_assert((BM / WM) * (BN / WN) == 4, "4 wavefronts per workgroup");
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
  constexpr int STAGE = A_BYTES + B_BYTES;
  constexpr int C_PITCH = (BN + 8) * 2;  // bytes per row of the bf16 C tile staged in LDS
  constexpr int LDS_BYTES = (NSTG * STAGE > BM * C_PITCH) ? NSTG * STAGE : BM * C_PITCH;
  __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm0 = (wave / WAVES_N) * WM;
  const int wn0 = (wave % WAVES_N) * WN;

  const int tiles_n = (g.N + BN - 1) / BN;
  const int tiles_m = (g.M + BM - 1) / BM;
  int tile_m, tile_n;
  if (g.m_fast) {
    tile_n = block_x / tiles_m;
    tile_m = block_x - tile_n * tiles_m;
  } else {
    tile_m = block_x / tiles_n;
    tile_n = block_x - tile_m * tiles_n;
  }
  const int gF = (EPI == EPI_BF16) ? g.epi.geglu_F : 0;  // GEGLU tiles: [80 value | 80 gate] columns
  const int m0 = tile_m * BM, n0 = tile_n * (gF ? BN / 2 : BN);

  // K range of this split; tiles past kt_end read as zeros (K limit folded into the loaders)
  const int kt1 = (m0 + BM <= g.seg1_row0) ? 0 : g.ktiles1;   // block-uniform: clean tiles of a twin batch skip segment 1
  const int kt_total = g.ktiles0 + kt1;
  const int kt_begin = (int)(((long)kt_total * block_z) / g.splits);
  const int kt_end = (int)(((long)kt_total * (block_z + 1)) / g.splits);
  const bool dual = kt1 > 0;

  DmaStager<BM, LA> sa;  // row descriptors only: tiles go global -> LDS directly (no staging registers, no ds_write)
  DmaStager<BN, LB> sb;
  constexpr int NLD = BM / 32 + BN / 32;  // LDS-DMA instructions per thread per K tile
  sa.begin(group_operand(g.a0, n0, g.grp_n, g.grp_a0), group_operand(g.a1, n0, g.grp_n, g.grp_a1), dual, m0, tid, kt_begin, kt_end, g.ktiles0);
  sb.begin(sample_operand(g.b0, m0), g.b1, dual, n0, tid, kt_begin, kt_end, g.ktiles0);

  uint2 biasr[FN];  // fetched ahead of the ring fill (older than every DMA: the counted vmcnt waits below retire them first)
  if constexpr (EPI == EPI_BF16) epi_load_bias<FN>(biasr, g.epi.bias, g.b0.base, n0, wn0, lane, g.N, gF, BN / 2);

  f32x4_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

  auto issue = [&](int stage, int) {  // stages the NEXT tile of the K range (tiles are requested in order)
    char* sA = lds + stage * STAGE;
    sa.dma(sA, wave);
    sb.dma(sA + A_BYTES, wave);
  };

  // NSTG LDS stages, NSTG-1 K tiles in flight; ONE barrier per K tile:
  //   wait own DMAs of tile t -> barrier (everybody's tile t landed, everybody finished reading tile t-1)
  //   -> refill tile t-1's stage with tile t+NSTG-1 -> MFMA on tile t.
  // Every step issues the same number of DMAs (tiles past the end are out of range: zero fill, no traffic), so the
  // counted wait is a compile-time constant.  The compiler does not order ds_read against LDS-DMA; the waits are manual.
#pragma unroll
  for (int u = 0; u < NSTG - 1; ++u) issue(u, kt_begin + u);

  int rd = 0, wr = NSTG - 1;
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    // ABL == 9: per-step timestamps of one wave (tools/trace_gemm.py); slots: loop top, data landed, barrier passed, DMA
    // issued; the next loop top closes the compute phase
    long* trace = nullptr;
    if (ABL == 9 && (block_x == 0 || block_x == 300) && block_z == 0 && (tid & 63) == 0)
      trace = reinterpret_cast<long*>(g.epi.Cf) + ((block_x ? 4 : 0) + wave) * 1024 + (kt - kt_begin) * 4;
    if (ABL == 9 && trace) trace[0] = clock64();
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTG - 2) * NLD) : "memory");
    if (ABL == 9 && trace) trace[1] = clock64();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (ABL == 9 && trace) trace[2] = clock64();
    if (ABL != 3 && ABL != 4) issue(wr, kt + NSTG - 1);
    if (ABL == 9 && trace) trace[3] = clock64();  // ABL: ablation probes (tools/tune_gemm.py), 0 in production
    const char* sA = lds + rd * STAGE;
    const char* sB = sA + A_BYTES;
#pragma unroll
    for (int ks = 0; ks < ((ABL == 2 || ABL == 4) ? 0 : BK / 32); ++ks) {
      bf16x8_t fa[FM], fb[FN];
      const int chunk = ks * 4 + (lane >> 4);
#pragma unroll
      for (int i = 0; i < FM; ++i)
        fa[i] = *reinterpret_cast<const bf16x8_t*>(sA + lds_off(wm0 + i * 16 + (lane & 15), chunk));
#pragma unroll
      for (int j = 0; j < FN; ++j)
        fb[j] = *reinterpret_cast<const bf16x8_t*>(sB + lds_off(wn0 + j * 16 + (lane & 15), chunk));
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          if (ABL == 1) {  // keep the LDS reads alive without the MFMA
            asm volatile("" ::"v"(fb[j]), "v"(fa[i]));
          } else {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
          }
    }
    rd = (rd + 1 == NSTG) ? 0 : rd + 1;
    wr = (wr + 1 == NSTG) ? 0 : wr + 1;
  }
  // the trailing (zero-fill) DMAs still write LDS: retire them before the C tile reuses it
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---------------------------------------------------------------- epilogue
  // acc[i][j][e]: output row m = m0+wm0+i*16+(lane&15); col n = n0+wn0+j*16 + 4*(lane>>4) + e
  const EpiParams& ep = g.epi;
  if constexpr (EPI == EPI_BF16) {
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
    if (gF) geglu_store<BM, BN, C_PITCH, NTHREADS>(lds, m0, n0, g.M, ep, tid);
    else epi_store_tile<BM, BN, C_PITCH, NTHREADS>(lds, m0, n0, g.M, g.N, ep, tid);
  } else {
    float* out = ep.Cf + (long)block_z * g.M * ep.ldcf;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int m = m0 + wm0 + i * 16 + (lane & 15);
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int n = n0 + wn0 + j * 16 + (lane >> 4) * 4;
        if (m >= g.M || n >= g.N) continue;
        *reinterpret_cast<float4*>(out + (long)m * ep.ldcf + n) =
            make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
      }
    }
  }
}

template <int BM, int BN, int WM, int WN, class LA, class LB, int EPI, int NSTG, int ABL = 0>
__global__ __launch_bounds__(NTHREADS) void gemm_kernel_d(const GemmArgs<LA, LB> g) {
  const int nblk = gridDim.x, bid = blockIdx.x;
  const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
  const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  gemm_body_d<BM, BN, WM, WN, LA, LB, EPI, NSTG, ABL>(g, logical, blockIdx.z);
}

template <int BM, int BN, int WM, int WN, class LA, class LB, int EPI, int NSTG, int ABL = 0>
inline void launch_gemm_d(const GemmArgs<LA, LB>& g, hipStream_t stream) {
  dim3 grid(aql_cdiv(g.M, BM) * aql_cdiv(g.N, BN), 1, g.splits);
  hipLaunchKernelGGL((gemm_kernel_d<BM, BN, WM, WN, LA, LB, EPI, NSTG, ABL>), grid, dim3(NTHREADS), 0, stream, g);
}

// NCW = compute wavefronts: 4 (8-wave workgroup) or 8 (12-wave workgroup, 256x160 tile: two compute wavefronts per SIMD keep the
// matrix pipe fed across each other's fragment-read latency, and a weight tile is shared by twice the rows).
// NLW = loader wavefronts: 4, or 8 (experiment, AQL_TILE=16 on a -DAQL_BIGWAVE build: the K loop is paced by the ISSUE of the LDS-DMA pieces,
// ~110 cycles per piece per loader wavefront -- twice the issuers, half the pieces each)
template <int BM, int BN, int WM, int WN, class LA, class LB, int EPI, int NSTG, int NCW = 4, int NLW = 4>
__device__ __forceinline__ void gemm_body_w(const GemmArgs<LA, LB>& g, const int block_x, const int block_z) {
  static_assert(EPI != EPI_ATOMIC && !LA::kTrans, "bf16 / slab epilogues only");
  constexpr int FM = WM / 16, FN = WN / 16;
  constexpr int WAVES_N = BN / WN;
  constexpr int NT = (NCW + NLW) * 64;   // threads per workgroup
  constexpr int RPI = 8 * NLW;           // rows per piece instruction across the loader wavefronts
  static_assert((BM / WM) * (BN / WN) == NCW && (NCW == 4 || NCW == 8), "NCW compute wavefronts (+ NLW loader wavefronts) per workgroup");
  static_assert(NLW == 4 || NLW == 8, "4 or 8 loader wavefronts");
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
  constexpr int STAGE = A_BYTES + B_BYTES;
  constexpr int C_PITCH = (BN + 8) * 2;  // bytes per row of the bf16 C tile staged in LDS
  constexpr int LDS_BYTES = (NSTG * STAGE > BM * C_PITCH) ? NSTG * STAGE : BM * C_PITCH;
  __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];

  // Wave specialisation: wavefronts 0-3 only read LDS and issue MFMAs, wavefronts 4-7 only issue the LDS-DMA loads.  An
  // in-kernel trace of the 4-wave kernel showed a wave stuck ~720 of 1800 cycles per K tile in the ISSUE of its loads
  // (the texture-address unit accepts 64 B/clk, the instruction blocks until accepted) before it could start its MFMAs;
  // with dedicated loader waves the matrix pipes never wait behind a load issue.
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: the role branch below is a uniform branch
  const bool loader = wave >= NCW;
  const int ltid = tid - NCW * 64;                              // loader waves: the 256-thread staging layout
  const int wm0 = ((wave % NCW) / WAVES_N) * WM;
  const int wn0 = ((wave % NCW) % WAVES_N) * WN;

  const int tiles_n = (g.N + BN - 1) / BN;
  const int tiles_m = (g.M + BM - 1) / BM;
  int tile_m, tile_n;
  if (g.m_fast) {
    tile_n = block_x / tiles_m;
    tile_m = block_x - tile_n * tiles_m;
  } else {
    tile_m = block_x / tiles_n;
    tile_n = block_x - tile_m * tiles_n;
  }
  const int gF = (EPI == EPI_BF16) ? g.epi.geglu_F : 0;  // GEGLU tiles: [80 value | 80 gate] columns
  const int m0 = tile_m * BM, n0 = tile_n * (gF ? BN / 2 : BN);

  // K range of this split; tiles past kt_end read as zeros (K limit folded into the loaders)
  const int kt1 = (m0 + BM <= g.seg1_row0) ? 0 : g.ktiles1;   // block-uniform: clean tiles of a twin batch skip segment 1
  const int kt_total = g.ktiles0 + kt1;
  const int kt_begin = (int)(((long)kt_total * block_z) / g.splits);
  const int kt_end = (int)(((long)kt_total * (block_z + 1)) / g.splits);
  const bool dual = kt1 > 0;

  constexpr int NLD = (BM + RPI - 1) / RPI + (BN + RPI - 1) / RPI;  // LDS-DMA instructions per loader thread per K tile (the wavefronts whose rows start inside the tile)
  constexpr int NLD_S = BM / RPI + BN / RPI;                       // ... of the wavefronts past a ragged tail (NLW = 8, BN = 160: 4 instead of 5)
  // this loader wavefront issues the short count when its 8 rows of the last piece lie past BM / BN (wave-uniform)
  const bool lw_short = loader && ((BM % RPI != 0 && (BM / RPI) * RPI + 8 * (wave - NCW) >= BM) || (BN % RPI != 0 && (BN / RPI) * RPI + 8 * (wave - NCW) >= BN));
  static_assert(NLD == NLD_S || NLD == NLD_S + 1, "at most one ragged operand");
  uint2 biasr[FN];  // compute wavefronts: this lane's bias values, fetched before the K loop (epi_load_bias)
  if constexpr (EPI == EPI_BF16) {
    if (!loader) epi_load_bias<FN>(biasr, g.epi.bias, g.b0.base, n0, wn0, lane, g.N, gF, BN / 2);
  }
  f32x4_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

  // Ring of NSTG (>= 3) LDS stages, ONE barrier per K tile, software-pipelined across tiles.  barrier(t) certifies that
  // tiles <= t+1 have landed and that nobody still reads tile t-1:
  //   loaders : issue t0..t0+NSTG-2 | wait t0 | B(pre) | { wait tile t+1 | B(t) | refill stage of t-1 with tile t+NSTG-1 }
  //   compute :                                B(pre) | read k-half 0 of t0 |
  //             { B(t) | read k-half 1 of t | MFMA k-half 0 of t | read k-half 0 of t+1 | MFMA k-half 1 of t }
  // so every fragment read is issued a full MFMA batch (20 x 16 cycles) before it is needed and nothing but the barrier
  // skew is exposed.  Tiles past the end are out-of-range DMAs (zero fill, no traffic): the counted waits are constants.
  // The two roles are separate loops (same barrier count) so that the stager state and the accumulators never share a
  // live range.
  static_assert(NSTG >= 3, "the cross-tile prefetch needs tile t+1 resident while tile t-1's stage is refilled");
  if (loader) {
    DmaStager<BM, LA, RPI> sa;  // row descriptors only: tiles go global -> LDS directly (no staging registers, no ds_write)
    DmaStager<BN, LB, RPI> sb;
    sa.begin(group_operand(g.a0, n0, g.grp_n, g.grp_a0), group_operand(g.a1, n0, g.grp_n, g.grp_a1), dual, m0, ltid, kt_begin, kt_end, g.ktiles0);
    sb.begin(sample_operand(g.b0, m0), g.b1, dual, n0, ltid, kt_begin, kt_end, g.ktiles0);
    auto issue = [&](int stage) {  // stages the NEXT tile of the K range (tiles are requested in order)
      char* sA = lds + stage * STAGE;
      sa.dma(sA, wave - NCW);
      sb.dma(sA + A_BYTES, wave - NCW);
    };
#pragma unroll
    for (int u = 0; u < NSTG - 1; ++u) issue(u);
    if (NLD != NLD_S && lw_short) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTG - 2) * NLD_S) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTG - 2) * NLD) : "memory");  // first tile landed
    __builtin_amdgcn_s_barrier();
    int wr = NSTG - 1;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
#ifdef AQL_TRACE_W
      long* trace = nullptr;
      if ((block_x == 0 || block_x == 100) && block_z == 0 && lane == 0)
        trace = reinterpret_cast<long*>(g.epi.Cf) + ((block_x ? 8 : 0) + wave) * 1024 + (kt - kt_begin) * 4;
      if (trace) trace[0] = clock64();
#endif
      if (NLD != NLD_S && lw_short) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTG - 3) * NLD_S) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTG - 3) * NLD) : "memory");  // tile kt+1 landed
#ifdef AQL_TRACE_W
      if (trace) trace[1] = clock64();
#endif
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
#ifdef AQL_TRACE_W
      if (trace) trace[2] = clock64();
#endif
      issue(wr);
#ifdef AQL_TRACE_W
      if (trace) trace[3] = clock64();
#endif
      wr = (wr + 1 == NSTG) ? 0 : wr + 1;
    }
  } else {
    const int arow = (wm0 + (lane & 15)), brow = (wn0 + (lane & 15));
    const int ch0 = lane >> 4, ch1 = 4 + (lane >> 4);
    if constexpr (FM >= 8) {
      // 128-row wavefront tiles (experiment, -DAQL_BIGWAVE: 256 x 160 on FOUR compute wavefronts): 40 accumulator fragments = 160
      // registers, so the A fragments cannot be held for a whole k-half (two halves x 8 fragments = 64 more registers: the
      // double-buffered loop below spilled 78 VGPRs).  Here the B fragments of both k-halves stay resident (40 registers) and the
      // A fragments stream through a ring of four: the read of fragment i+2 is issued in front of the five MFMAs of fragment i,
      // crossing k-halves and, at the end of a tile, into the next stage (tile t+1 has landed before barrier(t), as for the loop below).
      // Per 64-deep K step a wavefront reads 8 x 2 + 5 x 2 = 26 fragments for 80 MFMAs instead of 18 for 40.
      // MEASURED (profiles/r03_bigwave_experiment.txt): parity-green, 223-233 VGPRs, no spill -- and within 1 % of the 12-wave 256 x 160
      // kernel on every GEMM / conv shape tried.  Halving the fragment reads per MFMA changes nothing: the K loop of a 256 x 160 tile is
      // bound by the ISSUE of its 53 LDS-DMA pieces per step (4 loader wavefronts x 13 pieces x ~110 cycles), not by the LDS array.
      constexpr int NSUB = 2 * FM;            // (k-half, fragment) sub-steps per K tile
      bf16x8_t fbk[2][FN], far[4];
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
#pragma unroll
      for (int j = 0; j < FN; ++j) fbk[0][j] = *reinterpret_cast<const bf16x8_t*>(lds + A_BYTES + lds_off(brow + j * 16, ch0));
      far[0] = *reinterpret_cast<const bf16x8_t*>(lds + lds_off(arow, ch0));
      far[1] = *reinterpret_cast<const bf16x8_t*>(lds + lds_off(arow + 16, ch0));
      int rd = 0;
      for (int kt = kt_begin; kt < kt_end; ++kt) {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const char* sA = lds + rd * STAGE;
        const char* sB = sA + A_BYTES;
        rd = (rd + 1 == NSTG) ? 0 : rd + 1;
        const char* nA = lds + rd * STAGE;
        const char* nB = nA + A_BYTES;
#pragma unroll
        for (int u = 0; u < NSUB; ++u) {
          const int kh = u / FM, i = u % FM;
          // A fragment two sub-steps ahead: same tile (k-half 0 / 1) or fragments 0, 1 of the next tile's k-half 0
          const int v = u + 2;
          if (v < NSUB) far[v & 3] = *reinterpret_cast<const bf16x8_t*>(sA + lds_off(arow + (v % FM) * 16, (v / FM) ? ch1 : ch0));
          else far[v & 3] = *reinterpret_cast<const bf16x8_t*>(nA + lds_off(arow + (v - NSUB) * 16, ch0));
          // B fragments of the other k-half / the next tile, one per sub-step at the head of each half
          if (kh == 0 && i < FN) fbk[1][i] = *reinterpret_cast<const bf16x8_t*>(sB + lds_off(brow + i * 16, ch1));
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < FN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fbk[kh][j], far[u & 3], acc[i][j], 0, 0, 0);
          if (kh == 1 && i < FN) fbk[0][i] = *reinterpret_cast<const bf16x8_t*>(nB + lds_off(brow + i * 16, ch0));
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else {
    bf16x8_t fa0[FM], fb0[FN], fa1[FM], fb1[FN];
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int j = 0; j < FN; ++j) fb0[j] = *reinterpret_cast<const bf16x8_t*>(lds + A_BYTES + lds_off(brow + j * 16, ch0));
#pragma unroll
    for (int i = 0; i < FM; ++i) fa0[i] = *reinterpret_cast<const bf16x8_t*>(lds + lds_off(arow + i * 16, ch0));
    int rd = 0;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
#ifdef AQL_TRACE_W
      long* trace = nullptr;
      if ((block_x == 0 || block_x == 100) && block_z == 0 && lane == 0)
        trace = reinterpret_cast<long*>(g.epi.Cf) + ((block_x ? 8 : 0) + wave) * 1024 + (kt - kt_begin) * 4;
      if (trace) trace[0] = clock64();
#endif
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
#ifdef AQL_TRACE_W
      if (trace) trace[1] = trace[2] = clock64();
#endif
      const char* sA = lds + rd * STAGE;
      const char* sB = sA + A_BYTES;
      rd = (rd + 1 == NSTG) ? 0 : rd + 1;
      const char* nA = lds + rd * STAGE;
      const char* nB = nA + A_BYTES;
#pragma unroll
      for (int j = 0; j < FN; ++j) fb1[j] = *reinterpret_cast<const bf16x8_t*>(sB + lds_off(brow + j * 16, ch1));
#pragma unroll
      for (int i = 0; i < FM; ++i) fa1[i] = *reinterpret_cast<const bf16x8_t*>(sA + lds_off(arow + i * 16, ch1));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb0[j], fa0[i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < FN; ++j) fb0[j] = *reinterpret_cast<const bf16x8_t*>(nB + lds_off(brow + j * 16, ch0));
#pragma unroll
      for (int i = 0; i < FM; ++i) fa0[i] = *reinterpret_cast<const bf16x8_t*>(nA + lds_off(arow + i * 16, ch0));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb1[j], fa1[i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    }   // FM < 8
  }
  // the trailing (zero-fill) DMAs still write LDS: retire them before the C tile reuses it
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---------------------------------------------------------------- epilogue
  // acc[i][j][e]: output row m = m0+wm0+i*16+(lane&15); col n = n0+wn0+j*16 + 4*(lane>>4) + e
  const EpiParams& ep = g.epi;
  if constexpr (EPI == EPI_BF16) {
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
    if (gF) geglu_store<BM, BN, C_PITCH, NT>(lds, m0, n0, g.M, ep, tid);
    else epi_store_tile<BM, BN, C_PITCH, NT>(lds, m0, n0, g.M, g.N, ep, tid);
  } else if (!loader) {
    float* out = ep.Cf + (long)block_z * g.M * ep.ldcf;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int m = m0 + wm0 + i * 16 + (lane & 15);
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int n = n0 + wn0 + j * 16 + (lane >> 4) * 4;
        if (m >= g.M || n >= g.N) continue;
        *reinterpret_cast<float4*>(out + (long)m * ep.ldcf + n) =
            make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
      }
    }
  }
}

template <int BM, int BN, int WM, int WN, class LA, class LB, int EPI, int NSTG, int NCW = 4, int NLW = 4>
__global__ __launch_bounds__((NCW + NLW) * 64) void gemm_kernel_w(const GemmArgs<LA, LB> g) {
  const int nblk = gridDim.x, bid = blockIdx.x;
  const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
  const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  gemm_body_w<BM, BN, WM, WN, LA, LB, EPI, NSTG, NCW, NLW>(g, logical, blockIdx.z);
}

template <int BM, int BN, int WM, int WN, class LA, class LB, int EPI, int NSTG, int NCW = 4, int NLW = 4>
inline void launch_gemm_w(const GemmArgs<LA, LB>& g, hipStream_t stream) {
  dim3 grid(aql_cdiv(g.M, BM) * aql_cdiv(g.N, BN), 1, g.splits);
  hipLaunchKernelGGL((gemm_kernel_w<BM, BN, WM, WN, LA, LB, EPI, NSTG, NCW, NLW>), grid, dim3((NCW + NLW) * 64), 0, stream, g);
}

template <int BM, int BN, int WM, int WN, class LA, class LB, int EPI, int NSTG = 2>
__global__ __launch_bounds__(NTHREADS) void gemm_kernel(const GemmArgs<LA, LB> g) {
  // XCD-aware remap (guide T1): hardware block b runs on XCD b % 8, each XCD has a private L2.  Give every XCD a
  // CONTIGUOUS run of logical tiles so that tiles sharing an operand panel hit the same L2 (bijective for any grid).
  const int nblk = gridDim.x, bid = blockIdx.x;
  const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
  const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  gemm_body<BM, BN, WM, WN, LA, LB, EPI, NSTG>(g, logical, blockIdx.z);
}

// Grouped token-reduction GEMMs: ONE launch for all LoRA weight gradients of a backward pass.  Every problem has a
// narrow (rank <= 32) side, so all of them share the 128x32 tile; blockIdx.x is mapped to (problem, tile, K split)
// through the descriptor table (first_block is a running prefix).
struct TnGroupDesc {
  const bf16_t* a;   // [M][lda], rows of the output tile come from its columns (the wide side)
  const bf16_t* b;   // [M][ldb], the narrow side
  float* C;
  long lda, ldb, ldc;
  int M, a_rows, b_rows, splits;
  int first_block, trans_out;
  float alpha;
  int pad;
};

// `block_base` lets a launch cover a sub-range of the table (one gradient bucket of the data-parallel exchange): descs
// then points at the bucket's first descriptor and block_base is that descriptor's first_block.
static __global__ __launch_bounds__(NTHREADS) void gemm_tn_grouped_kernel(const TnGroupDesc* __restrict__ descs, int n,
                                                                   int block_base) {
  const int bid = (int)blockIdx.x + block_base;
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (descs[mid].first_block <= bid) lo = mid; else hi = mid - 1;
  }
  const TnGroupDesc d = descs[lo];
  const int local = bid - d.first_block;
  const int tiles = (d.a_rows + 127) / 128;  // b_rows <= 32: one tile along N
  GemmArgs<TransLoader, TransLoader> g;
  g.a0.base = d.a; g.a0.ld = d.lda; g.a0.rows = d.a_rows; g.a0.K = d.M;
  g.b0.base = d.b; g.b0.ld = d.ldb; g.b0.rows = d.b_rows; g.b0.K = d.M;
  g.a1 = g.a0; g.b1 = g.b0;
  g.ktiles0 = (d.M + BK - 1) / BK; g.ktiles1 = 0;
  g.M = d.a_rows; g.N = d.b_rows; g.splits = d.splits; g.m_fast = 0;
  g.epi = EpiParams{};
  g.epi.Cf = d.C; g.epi.ldcf = d.ldc; g.epi.alpha = d.alpha; g.epi.trans_out = d.trans_out;
  gemm_body<128, 32, 32, 32, TransLoader, TransLoader, EPI_ATOMIC>(g, local % tiles, local / tiles);
}

template <int BM, int BN, int WM, int WN, class LA, class LB, int EPI, int NSTG = 2>
inline void launch_gemm(const GemmArgs<LA, LB>& g, hipStream_t stream) {
  dim3 grid(aql_cdiv(g.M, BM) * aql_cdiv(g.N, BN), 1, g.splits);
  hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, LA, LB, EPI, NSTG>), grid, dim3(NTHREADS), 0, stream, g);
}

}  // namespace aqlgemm
