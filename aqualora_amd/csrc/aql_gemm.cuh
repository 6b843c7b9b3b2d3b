// bf16 MFMA GEMM core for gfx950 (CDNA4): one templated main loop shared by
//   * the fused LoRA linear  Y = X.W^T + b + (T*S).Bup^T      (two K segments, one accumulator)
//   * the 3x3 NHWC convolution as an implicit GEMM (gather loader, nearest-x2 upsample and stride folded in)
//   * its backward-data form, and
//   * the weight-gradient GEMMs dA/dB (reduction over tokens, transposing loader).
//
// Tile: BM x BN outputs per 256-thread workgroup (4 wavefronts of 64), K tile 64 bf16, double-buffered LDS,
// v_mfma_f32_32x32x16_bf16 with the *weight-side* tile as MFMA operand A and the *activation-side* tile as
// operand B, so a lane ends up with 4 consecutive output columns of one output row (8-byte packed stores).
// LDS rows are 128 B (8 chunks of 16 B); chunk index is XOR-swizzled with (row & 7) so that both the 16-byte
// staging writes and the ds_read_b128 fragment reads are bank-conflict free (guide T2).
#pragma once
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
  struct Row {
    const bf16_t* p;
  };
  struct KInfo {
    int k;
  };
  __device__ __forceinline__ Row row(int r) const {
    Row x;
    x.p = (r < rows) ? base + (long)r * ld : nullptr;
    return x;
  }
  __device__ __forceinline__ KInfo kinfo(int k) const { return KInfo{k < K ? k : -1}; }
  __device__ __forceinline__ uint4 load(const Row& x, const KInfo& ki) const {
    if (x.p != nullptr && ki.k >= 0) return *reinterpret_cast<const uint4*>(x.p + ki.k);
    return zero4();
  }
};

// Forward 3x3 conv, NHWC input [B,Hin,Win,Cin]; row r = (b,ho,wo); k = (kh*3+kw)*Cin + ci.
// ups=1 reads the input through a nearest x2 upsample (diffusers Upsample2D) without materialising it.
struct ConvFwdLoader {
  static constexpr bool kTrans = false;
  const bf16_t* base;
  int B, Hin, Win, Cin, Hout, Wout, stride, ups;
  int rows, K;
  struct Row {
    int b, h, w;
  };
  struct KInfo {
    int kh, kw, ci;
  };
  __device__ __forceinline__ Row row(int r) const {
    Row x;
    if (r >= rows) {
      x.b = -1;
      x.h = x.w = 0;
      return x;
    }
    int hw = Hout * Wout;
    x.b = r / hw;
    int rem = r - x.b * hw;
    int ho = rem / Wout;
    x.h = ho * stride - 1;
    x.w = (rem - ho * Wout) * stride - 1;
    return x;
  }
  __device__ __forceinline__ KInfo kinfo(int k) const {
    KInfo ki;
    if (k >= K) {
      ki.kh = -100;
      ki.kw = 0;
      ki.ci = 0;
      return ki;
    }
    int tap = k / Cin;
    ki.ci = k - tap * Cin;
    ki.kh = tap / 3;
    ki.kw = tap - ki.kh * 3;
    return ki;
  }
  __device__ __forceinline__ uint4 load(const Row& x, const KInfo& ki) const {
    int hi = x.h + ki.kh, wi = x.w + ki.kw;
    int Hl = Hin << ups, Wl = Win << ups;
    if (x.b < 0 || hi < 0 || wi < 0 || hi >= Hl || wi >= Wl) return zero4();
    hi >>= ups;
    wi >>= ups;
    return *reinterpret_cast<const uint4*>(base + (((long)x.b * Hin + hi) * Win + wi) * Cin + ki.ci);
  }
};

// Backward-data 3x3 conv: rows are input pixels (b,hi,wi) of dX; source is dY [B,Hout,Wout,Cout];
// k = (kh*3+kw)*Cout + co;  hi = ho*stride + kh - 1.
struct ConvBwdLoader {
  static constexpr bool kTrans = false;
  const bf16_t* base;
  int B, Hin, Win, Cout, Hout, Wout, stride;
  int rows, K;
  struct Row {
    int b, h, w;
  };
  struct KInfo {
    int kh, kw, co;
  };
  __device__ __forceinline__ Row row(int r) const {
    Row x;
    if (r >= rows) {
      x.b = -1;
      x.h = x.w = 0;
      return x;
    }
    int hw = Hin * Win;
    x.b = r / hw;
    int rem = r - x.b * hw;
    int hi = rem / Win;
    x.h = hi + 1;
    x.w = (rem - hi * Win) + 1;
    return x;
  }
  __device__ __forceinline__ KInfo kinfo(int k) const {
    KInfo ki;
    if (k >= K) {
      ki.kh = 100000;
      ki.kw = 0;
      ki.co = 0;
      return ki;
    }
    int tap = k / Cout;
    ki.co = k - tap * Cout;
    ki.kh = tap / 3;
    ki.kw = tap - ki.kh * 3;
    return ki;
  }
  __device__ __forceinline__ uint4 load(const Row& x, const KInfo& ki) const {
    int th = x.h - ki.kh, tw = x.w - ki.kw;
    if (x.b < 0 || th < 0 || tw < 0) return zero4();
    if (stride == 2) {
      if ((th | tw) & 1) return zero4();
      th >>= 1;
      tw >>= 1;
    }
    if (th >= Hout || tw >= Wout) return zero4();
    return *reinterpret_cast<const uint4*>(base + (((long)x.b * Hout + th) * Wout + tw) * Cout + ki.co);
  }
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
  bf16_t* C2;              // second output (row-scaled) or null
  long ldc2;
  const bf16_t* rowscale;  // [nsamples][N]
  const bf16_t* rowbias;   // [nsamples][rowbias_ld] added (bf16 add) after bias, before the residual; or null
  long rowbias_ld;
  int rows_per_sample;
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
  int splits;  // grid.z; k tiles of the concatenated K range are divided evenly
  int m_fast;  // tile order: 0 = N tiles of one M tile adjacent (activation reuse), 1 = M tiles adjacent (weight reuse)
  EpiParams epi;
};

// --------------------------------------------------------------------------------------------
// staging helpers
// --------------------------------------------------------------------------------------------
template <int R, class L>
struct Stager {
  // non-transposed: thread owns chunk c = tid&7 of rows (tid>>3) + 32*i
  static constexpr int NL = R / 32;
  typename L::Row rows[NL];
  uint4 regs[NL];
  __device__ __forceinline__ void init(const L& l, int row0, int tid) {
#pragma unroll
    for (int i = 0; i < NL; ++i) rows[i] = l.row(row0 + (tid >> 3) + 32 * i);
  }
  __device__ __forceinline__ void fetch(const L& l, int k0, int tid) {
    typename L::KInfo ki = l.kinfo(k0 + (tid & 7) * 8);
#pragma unroll
    for (int i = 0; i < NL; ++i) regs[i] = l.load(rows[i], ki);
  }
  __device__ __forceinline__ void commit(char* lds, int tid) const {
#pragma unroll
    for (int i = 0; i < NL; ++i)
      *reinterpret_cast<uint4*>(lds + lds_off((tid >> 3) + 32 * i, tid & 7)) = regs[i];
  }
};

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
// Pipelined main loop (non-transposed loaders).  The plain double-buffered loop above keeps ONE K tile of global
// loads in flight per workgroup, and with <= 2 workgroups per CU a K tile's MFMA work (a few hundred cycles) cannot
// cover an HBM / L2 round trip (1-2 k cycles): measured, every GEMM of the train step sat at 10-20 % MFMA use.  Here
// PD K tiles are in flight in REGISTERS (the 512 KB register file is the roomy resource, not the 160 KB LDS):
//   * operands are read with raw BUFFER loads: out-of-range rows / K tails / conv padding become an out-of-range
//     offset that the hardware answers with zeros - no branch around the load, so the compiler can count
//     (s_waitcnt vmcnt(n)) instead of draining the queue;
//   * 16x16x32 MFMA fragments, so BN = 160 tiles exist (every channel count of this U-Net is a multiple of 160).
// ------------------------------------------------------------------------------------------------
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
constexpr uint32_t OOB_ROW = 0x80000000u;  // any offset >= BUF_BYTES reads as zero
constexpr uint32_t OOB_K = 0x40000000u;
constexpr uint32_t BUF_BYTES = 0x40000000u;  // operands must be < 1 GiB (host-checked)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, p ? BUF_BYTES : 0u, 0x00020000);
}
__device__ __forceinline__ u32x4_t buf_load16(__amdgpu_buffer_rsrc_t r, uint32_t off) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
}

// PStager<R, L, PD>: thread owns 16-byte chunk c = tid&7 of rows (tid>>3) + 32*i of an R-row operand tile and keeps PD
// K tiles of them in registers.
template <int R, class L, int PD>
struct PStager;

template <int R, int PD>
struct PStager<R, PlainLoader, PD> {
  static constexpr int NL = R / 32;
  uint32_t rowoff0[NL], rowoff1[NL];
  __amdgpu_buffer_rsrc_t rs0, rs1;
  int klim0, klim1;
  u32x4_t regs[PD][NL];
  __device__ __forceinline__ void init(const PlainLoader& l0, const PlainLoader& l1, bool dual, int row0, int tid,
                                       int kend0, int kend1) {
    rs0 = make_rsrc(l0.base);
    rs1 = make_rsrc(dual ? l1.base : nullptr);
    klim0 = l0.K < kend0 ? l0.K : kend0;
    klim1 = dual ? (l1.K < kend1 ? l1.K : kend1) : 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int r = row0 + (tid >> 3) + 32 * i;
      rowoff0[i] = r < l0.rows ? (uint32_t)r * (uint32_t)(l0.ld * 2) : OOB_ROW;
      rowoff1[i] = (dual && r < l1.rows) ? (uint32_t)r * (uint32_t)(l1.ld * 2) : OOB_ROW;
    }
  }
  __device__ __forceinline__ void fetch(int slot, bool seg, int k) {
    const __amdgpu_buffer_rsrc_t rs = seg ? rs1 : rs0;
    const uint32_t koff = (k < (seg ? klim1 : klim0)) ? (uint32_t)k * 2u : OOB_K;
#pragma unroll
    for (int i = 0; i < NL; ++i) regs[slot][i] = buf_load16(rs, (seg ? rowoff1[i] : rowoff0[i]) + koff);
  }
  // LDS-DMA variant: the wave's 64 lanes fill 8 consecutive 128-B LDS rows; the XOR swizzle is applied on the SOURCE
  // side (lane (row, slot) fetches chunk slot ^ ((row >> 1) & 7)), kc = that chunk's first k
  __device__ __forceinline__ void dma(char* stage, int wave, bool seg, int k) const {
    const __amdgpu_buffer_rsrc_t rs = seg ? rs1 : rs0;
    const uint32_t koff = (k < (seg ? klim1 : klim0)) ? (uint32_t)k * 2u : OOB_K;
#pragma unroll
    for (int i = 0; i < NL; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(stage + (32 * i + 8 * wave) * 128), 16,
                                               (seg ? rowoff1[i] : rowoff0[i]) + koff, 0, 0, 0);
  }
  __device__ __forceinline__ void commit(int slot, char* lds, int tid) const {
#pragma unroll
    for (int i = 0; i < NL; ++i)
      *reinterpret_cast<u32x4_t*>(lds + lds_off((tid >> 3) + 32 * i, tid & 7)) = regs[slot][i];
  }
};

template <int R, int PD>
struct PStager<R, ConvFwdLoader, PD> {
  static constexpr int NL = R / 32;
  int rb[NL], rh[NL], rw[NL];  // b*Hin, h0, w0 of the row's output pixel (h0 = -2^20 for rows past M)
  __amdgpu_buffer_rsrc_t rs;
  int klim, Cin, Win, Hl, Wl, ups;
  u32x4_t regs[PD][NL];
  __device__ __forceinline__ void init(const ConvFwdLoader& l, const ConvFwdLoader&, bool, int row0, int tid, int kend0,
                                       int) {
    rs = make_rsrc(l.base);
    klim = l.K < kend0 ? l.K : kend0;
    Cin = l.Cin;
    Win = l.Win;
    ups = l.ups;
    Hl = l.Hin << l.ups;
    Wl = l.Win << l.ups;
    const int hw = l.Hout * l.Wout;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int r = row0 + (tid >> 3) + 32 * i;
      const int b = r / hw, rem = r - b * hw, ho = rem / l.Wout;
      rb[i] = b * l.Hin;
      rh[i] = r < l.rows ? ho * l.stride - 1 : -(1 << 20);
      rw[i] = (rem - ho * l.Wout) * l.stride - 1;
    }
  }
  __device__ __forceinline__ void fetch(int slot, bool seg, int k) {
    const bool kok = !seg && k < klim;
    const int tap = k / Cin, ci = k - tap * Cin;
    const int kh = kok ? tap / 3 : -(1 << 20), kw = tap - (tap / 3) * 3;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int hi = rh[i] + kh, wi = rw[i] + kw;
      const bool ok = (unsigned)hi < (unsigned)Hl && (unsigned)wi < (unsigned)Wl;
      const uint32_t off = (uint32_t)(((rb[i] + (hi >> ups)) * Win + (wi >> ups)) * Cin + ci) * 2u;
      regs[slot][i] = buf_load16(rs, ok ? off : OOB_ROW);
    }
  }
  __device__ __forceinline__ void dma(char* stage, int wave, bool seg, int k) const {
    const bool kok = !seg && k < klim;
    const int tap = k / Cin, ci = k - tap * Cin;
    const int kh = kok ? tap / 3 : -(1 << 20), kw = tap - (tap / 3) * 3;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int hi = rh[i] + kh, wi = rw[i] + kw;
      const bool ok = (unsigned)hi < (unsigned)Hl && (unsigned)wi < (unsigned)Wl;
      const uint32_t off = (uint32_t)(((rb[i] + (hi >> ups)) * Win + (wi >> ups)) * Cin + ci) * 2u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(stage + (32 * i + 8 * wave) * 128), 16,
                                               ok ? off : OOB_ROW, 0, 0, 0);
    }
  }
  __device__ __forceinline__ void commit(int slot, char* lds, int tid) const {
#pragma unroll
    for (int i = 0; i < NL; ++i)
      *reinterpret_cast<u32x4_t*>(lds + lds_off((tid >> 3) + 32 * i, tid & 7)) = regs[slot][i];
  }
};

template <int R, int PD>
struct PStager<R, ConvBwdLoader, PD> {
  static constexpr int NL = R / 32;
  int rb[NL], rh[NL], rw[NL];  // b*Hout, hi+1, wi+1 of the row's input pixel
  __amdgpu_buffer_rsrc_t rs;
  int klim, Cout, Hout, Wout, s2;
  u32x4_t regs[PD][NL];
  __device__ __forceinline__ void init(const ConvBwdLoader& l, const ConvBwdLoader&, bool, int row0, int tid, int kend0,
                                       int) {
    rs = make_rsrc(l.base);
    klim = l.K < kend0 ? l.K : kend0;
    Cout = l.Cout;
    Hout = l.Hout;
    Wout = l.Wout;
    s2 = l.stride == 2 ? 1 : 0;
    const int hw = l.Hin * l.Win;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int r = row0 + (tid >> 3) + 32 * i;
      const int b = r / hw, rem = r - b * hw, hi = rem / l.Win;
      rb[i] = b * l.Hout;
      rh[i] = r < l.rows ? hi + 1 : -(1 << 20);
      rw[i] = (rem - hi * l.Win) + 1;
    }
  }
  __device__ __forceinline__ void fetch(int slot, bool seg, int k) {
    const bool kok = !seg && k < klim;
    const int tap = k / Cout, co = k - tap * Cout;
    const int kh = kok ? tap / 3 : (1 << 21), kw = tap - (tap / 3) * 3;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int th = rh[i] - kh, tw = rw[i] - kw;
      const bool par = ((th | tw) & s2) == 0;  // stride 2: only even offsets hit an output pixel
      const int ho = th >> s2, wo = tw >> s2;
      const bool ok = par && th >= 0 && tw >= 0 && ho < Hout && wo < Wout;
      const uint32_t off = (uint32_t)(((rb[i] + ho) * Wout + wo) * Cout + co) * 2u;
      regs[slot][i] = buf_load16(rs, ok ? off : OOB_ROW);
    }
  }
  __device__ __forceinline__ void dma(char* stage, int wave, bool seg, int k) const {
    const bool kok = !seg && k < klim;
    const int tap = k / Cout, co = k - tap * Cout;
    const int kh = kok ? tap / 3 : (1 << 21), kw = tap - (tap / 3) * 3;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int th = rh[i] - kh, tw = rw[i] - kw;
      const bool par = ((th | tw) & s2) == 0;
      const int ho = th >> s2, wo = tw >> s2;
      const bool ok = par && th >= 0 && tw >= 0 && ho < Hout && wo < Wout;
      const uint32_t off = (uint32_t)(((rb[i] + ho) * Wout + wo) * Cout + co) * 2u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(stage + (32 * i + 8 * wave) * 128), 16,
                                               ok ? off : OOB_ROW, 0, 0, 0);
    }
  }
  __device__ __forceinline__ void commit(int slot, char* lds, int tid) const {
#pragma unroll
    for (int i = 0; i < NL; ++i)
      *reinterpret_cast<u32x4_t*>(lds + lds_off((tid >> 3) + 32 * i, tid & 7)) = regs[slot][i];
  }
};

template <int BM, int BN, int WM, int WN, class LA, class LB, int EPI, int PD>
__device__ __forceinline__ void gemm_body_p(const GemmArgs<LA, LB>& g, const int block_x, const int block_z) {
  static_assert(EPI != EPI_ATOMIC && !LA::kTrans, "bf16 / slab epilogues only");
  constexpr int FM = WM / 16, FN = WN / 16;
  constexpr int WAVES_N = BN / WN;
  static_assert((BM / WM) * (BN / WN) == 4, "4 wavefronts per workgroup");
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
  constexpr int STAGE = A_BYTES + B_BYTES;
  constexpr int C_PITCH = (BN + 8) * 2;  // bytes per row of the bf16 C tile staged in LDS
  constexpr int LDS_BYTES = (2 * STAGE > BM * C_PITCH) ? 2 * STAGE : BM * C_PITCH;
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

  // K range of this split; tiles past kt_end read as zeros (K limit folded into the loaders)
  const int kt_total = g.ktiles0 + g.ktiles1;
  const int kt_begin = (int)(((long)kt_total * block_z) / g.splits);
  const int kt_end = (int)(((long)kt_total * (block_z + 1)) / g.splits);
  const bool dual = g.ktiles1 > 0;

  PStager<BM, LA, PD> sa;
  PStager<BN, LB, PD> sb;
  sa.init(g.a0, g.a1, dual, m0, tid, kt_end * BK, (kt_end - g.ktiles0) * BK);
  sb.init(g.b0, g.b1, dual, n0, tid, kt_end * BK, (kt_end - g.ktiles0) * BK);

  f32x4_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

  auto issue = [&](int slot, int t) {
    const bool seg = t >= g.ktiles0;  // uniform
    const int k = (seg ? t - g.ktiles0 : t) * BK + (tid & 7) * 8;
    sa.fetch(slot, seg, k);
    sb.fetch(slot, seg, k);
  };

#pragma unroll
  for (int u = 0; u < PD; ++u) issue(u, kt_begin + u);
  sa.commit(0, lds, tid);
  sb.commit(0, lds + A_BYTES, tid);
  __syncthreads();

  int cur = 0;
  for (int kt = kt_begin; kt < kt_end; kt += PD) {
#pragma unroll
    for (int u = 0; u < PD; ++u) {
      const int t = kt + u;
      // LDS[cur] holds tile t (it came from slot u); refill the slot with tile t + PD.  Every step issues the same
      // loads whether or not t is past the end (then they are out of range and free), so the in-flight count is static.
      issue(u, t + PD);
      if (t < kt_end) {
        const char* sA = lds + cur * STAGE;
        const char* sB = sA + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < BK / 32; ++ks) {
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
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
      }
      // tile t + 1 (slot u+1) -> the other LDS stage; PD - 1 younger tiles stay in flight
      char* nA = lds + (cur ^ 1) * STAGE;
      sa.commit((u + 1) % PD, nA, tid);
      sb.commit((u + 1) % PD, nA + A_BYTES, tid);
      __syncthreads();
      cur ^= 1;
    }
  }

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
        if (ep.bias != nullptr && (n0 + col) < g.N) {
          const uint2 bb = *reinterpret_cast<const uint2*>(ep.bias + n0 + col);
          v0 += bf16lo(bb.x);
          v1 += bf16hi(bb.x);
          v2 += bf16lo(bb.y);
          v3 += bf16hi(bb.y);
        }
        *reinterpret_cast<uint2*>(lds + row * C_PITCH + col * 2) = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
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

template <int BM, int BN, int WM, int WN, class LA, class LB, int EPI, int PD>
__global__ __launch_bounds__(NTHREADS) void gemm_kernel_p(const GemmArgs<LA, LB> g) {
  const int nblk = gridDim.x, bid = blockIdx.x;
  const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
  const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  gemm_body_p<BM, BN, WM, WN, LA, LB, EPI, PD>(g, logical, blockIdx.z);
}

template <int BM, int BN, int WM, int WN, class LA, class LB, int EPI, int PD>
inline void launch_gemm_p(const GemmArgs<LA, LB>& g, hipStream_t stream) {
  dim3 grid(aql_cdiv(g.M, BM) * aql_cdiv(g.N, BN), 1, g.splits);
  hipLaunchKernelGGL((gemm_kernel_p<BM, BN, WM, WN, LA, LB, EPI, PD>), grid, dim3(NTHREADS), 0, stream, g);
}

template <int BM, int BN, int WM, int WN, class LA, class LB, int EPI, int NSTG>
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
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // K range of this split; tiles past kt_end read as zeros (K limit folded into the loaders)
  const int kt_total = g.ktiles0 + g.ktiles1;
  const int kt_begin = (int)(((long)kt_total * block_z) / g.splits);
  const int kt_end = (int)(((long)kt_total * (block_z + 1)) / g.splits);
  const bool dual = g.ktiles1 > 0;

  PStager<BM, LA, 1> sa;  // row descriptors only: tiles go global -> LDS directly (no staging registers, no ds_write)
  PStager<BN, LB, 1> sb;
  constexpr int NLD = BM / 32 + BN / 32;  // LDS-DMA instructions per thread per K tile
  sa.init(g.a0, g.a1, dual, m0, tid, kt_end * BK, (kt_end - g.ktiles0) * BK);
  sb.init(g.b0, g.b1, dual, n0, tid, kt_end * BK, (kt_end - g.ktiles0) * BK);

  f32x4_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

  // lane (row = 32i + tid>>3, slot = tid&7) fetches source chunk slot ^ ((row >> 1) & 7) = (tid&7) ^ ((tid>>4)&7)
  const int kc = (((tid & 7) ^ ((tid >> 4) & 7))) * 8;
  auto issue = [&](int stage, int t) {
    const bool seg = t >= g.ktiles0;  // uniform
    const int k = (seg ? t - g.ktiles0 : t) * BK + kc;
    char* sA = lds + stage * STAGE;
    sa.dma(sA, wave, seg, k);
    sb.dma(sA + A_BYTES, wave, seg, k);
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
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTG - 2) * NLD) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue(wr, kt + NSTG - 1);
    const char* sA = lds + rd * STAGE;
    const char* sB = sA + A_BYTES;
#pragma unroll
    for (int ks = 0; ks < BK / 32; ++ks) {
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
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
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
        if (ep.bias != nullptr && (n0 + col) < g.N) {
          const uint2 bb = *reinterpret_cast<const uint2*>(ep.bias + n0 + col);
          v0 += bf16lo(bb.x);
          v1 += bf16hi(bb.x);
          v2 += bf16lo(bb.y);
          v3 += bf16hi(bb.y);
        }
        *reinterpret_cast<uint2*>(lds + row * C_PITCH + col * 2) = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
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

template <int BM, int BN, int WM, int WN, class LA, class LB, int EPI, int NSTG>
__global__ __launch_bounds__(NTHREADS) void gemm_kernel_d(const GemmArgs<LA, LB> g) {
  const int nblk = gridDim.x, bid = blockIdx.x;
  const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
  const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  gemm_body_d<BM, BN, WM, WN, LA, LB, EPI, NSTG>(g, logical, blockIdx.z);
}

template <int BM, int BN, int WM, int WN, class LA, class LB, int EPI, int NSTG>
inline void launch_gemm_d(const GemmArgs<LA, LB>& g, hipStream_t stream) {
  dim3 grid(aql_cdiv(g.M, BM) * aql_cdiv(g.N, BN), 1, g.splits);
  hipLaunchKernelGGL((gemm_kernel_d<BM, BN, WM, WN, LA, LB, EPI, NSTG>), grid, dim3(NTHREADS), 0, stream, g);
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

__global__ __launch_bounds__(NTHREADS) void gemm_tn_grouped_kernel(const TnGroupDesc* __restrict__ descs, int n) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (descs[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const TnGroupDesc d = descs[lo];
  const int local = blockIdx.x - d.first_block;
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
