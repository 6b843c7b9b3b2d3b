// Row-resident chains of the transformer block at the 320-channel level (round 5).
//
// Reference graph: BasicTransformerBlock / Transformer2DModel of scripts/lib/original_unet.py:754-806, 809-890 with the watermark-LoRA
// linears of utils/lora_modules.py:9-26, 56-62 on every projection.  At 64 x 64 latents every linear of the block except the
// feed-forward pair has K = 320 and N = 320, and between two of them sit only row-local operations (bias, residual add, LayerNorm).
// Run one launch per linear, each of those is a 13-22 us kernel that moves 2 x 21 MB through HBM for 2.7 us of MFMA work (round 4:
// 0.28 of the HBM roof, MfmaUtil 13 %), and the LayerNorm between them re-reads and re-writes the same 21 MB.
//
// Here a workgroup OWNS 128 token rows for a whole chain of stages.  The 128 x 320 bf16 activation tile (80 KB) lives in LDS as the
// complete A panel of the next linear; the weights of the stage stream through a 3-stage LDS-DMA ring (K tiles of 32: 320 x 64 B of W
// + 32 x 64 B of the LoRA down matrix per tile); a stage's output tile is rounded to bf16 exactly where the one-launch kernels round
// it and either leaves for HBM straight from the accumulator registers (DIRECT: q | k | v, the last linear of a chain) or overwrites
// the resident tile (KEEP), where a ROW PASS adds the residual, stores the residual stream, runs LayerNorm (the arithmetic of
// ln_kernel, aql_norm.hip, operation for operation) and stores the normalised rows the backward pass needs.
//
//   stage g:   Y = X.W_g^T  [+ ((X.A_g^T) * S[sample]).Bup_g^T]  + bias_g              (fp32 accumulate; X = the resident tile)
//              KEEP:   X <- bf16(Y);  row pass:  X <- X + res_g (bf16 add);  out_g <- X;  [ X <- LayerNorm(X); nout_g <- X ]
//              DIRECT: out_g <- bf16(Y)
//
// Same operations in the same order as lora_gemm_kernel + ln_kernel: BIT-IDENTICAL outputs (tools/probe_chain.py,
// tests/test_gpu_kernels.py).  Twin batches (ops._Dual): tiles that lie below row0 (the clean half, all-zero scale rows) skip the
// LoRA side product, the up-projection step, T / Ts and the nout rows.
//
// LDS (161,600 of 163,840 bytes): resident tile 10 K tiles x 128 rows x 64 B | ring 3 x (20 KB + 2 KB) | Ts tile 128 x 64 B |
// bias of up to 4 stages, gamma, beta, the tile's scale row.  64-byte rows, 16-byte chunk index XOR g[(row >> 2) & 3] with
// g = {0, 2, 3, 1}: conflict-free for the ds_read_b128 fragment reads of v_mfma_f32_16x16x32_bf16 (hardware lane groups
// {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, +32: each group touches every 16-byte slot of a 256-byte bank window once).
#include "aql_gemm.cuh"
#include <stdlib.h>

using namespace aqlgemm;

namespace aqlchain {

constexpr int CH = 320, NTH = 512, NKT = CH / 32, LR = 32, MAXS = 4;
constexpr int W_BYTES = CH * 64, L_BYTES = LR * 64, STAGE = W_BYTES + L_BYTES, NSTG = 3;
// LDS map of a workgroup that owns BM = 64 FM rows (FM = 2: the forward chains on twin batches, 128-row tiles; FM = 1: the backward
// chains, whose 16384 rows would fill only half the chip with 128-row tiles)
template <int FM>
struct Lay {
  static constexpr int BM = 64 * FM;
  static constexpr int RES_BYTES = NKT * BM * 64;
  static constexpr int OFF_RING = RES_BYTES;
  static constexpr int OFF_TS = OFF_RING + NSTG * STAGE;
  static constexpr int OFF_BIAS = OFF_TS + BM * 64;
  static constexpr int OFF_GAMMA = OFF_BIAS + MAXS * CH * 2;
  static constexpr int OFF_BETA = OFF_GAMMA + CH * 2;
  static constexpr int OFF_SROW = OFF_BETA + CH * 2;
  static constexpr int TOTAL = OFF_SROW + 64;
};
static_assert(Lay<2>::TOTAL <= 160 * 1024, "LDS budget");

enum { RP_NONE = 0, RP_LN_FWD = 1, RP_LN_BWD = 2 };

struct Stage {
  const bf16_t *W, *bias, *Ad, *Bup;   // W [320][ldw], bias [320] or null, Ad [32][320] or null (no LoRA), Bup [320][32]
  bf16_t *T, *Ts;                      // [M][32], rows >= row0
  const bf16_t* res;                   // KEEP: rows [M][ldr] added (bf16) to the tile -- the residual; LN backward: the gradient of the residual branch, added last
  bf16_t* out;                         // [M][ldo] or null
  bf16_t* nout;                        // KEEP + forward LN: LayerNorm output [M][ldn] (rows >= nout_row0) or null
  float* stats;                        // forward LN: (mean, rstd) per row [M][2], written; LN backward: the saved statistics, read
  const bf16_t *gamma, *beta;
  const bf16_t* lnx;                   // LN backward: the LayerNorm's saved input rows [M][ldlx]
  long ldw, ldr, ldo, ldn, ldlx;
  float eps;
  float oscale;       // DIRECT stages: out = bf16((acc + bias) * oscale) -- 1, or scale log2(e) on attn1.to_q (aql_sdpa_*_qpre reads q pre-multiplied)
  int keep, ln, nout_row0;             // ln: RP_NONE / RP_LN_FWD / RP_LN_BWD
};

struct Args {
  const bf16_t* X;   // [M][ldx] chain input
  long ldx;
  const bf16_t* S;   // [nsamples][32] scale rows
  int M, rps, row0, nstage;
  int has_pre;        // a row pass on the chain INPUT in front of the first linear (backward chains: LN backward of the incoming gradient)
  long long* trace;   // -DAQL_CHAIN_TRACE builds (tools/trace_chain.py): 32 cycle stamps per (block, wave 0 / wave 4); null otherwise
  Stage pre;          // row-pass fields only
  Stage st[MAXS];
};

__device__ __forceinline__ int swz4(int r) { return (0x78 >> (((r >> 2) & 3) * 2)) & 3; }
__device__ __forceinline__ int off64(int row, int chunk) { return row * 64 + ((chunk ^ swz4(row)) << 4); }

// Sums of EIGHT rows across the wavefront in the operand order of wave_sum (aql_common.h: xor-32, xor-16, then row_ror 8 / 4 / 2 / 1
// inside a 16-lane row; IEEE adds are commutative, so every partial sum has the same bits), but transposed: a v_permlane32_swap of
// two rows' registers puts row A's two halves side by side in lanes 0-31 and row B's in lanes 32-63, so ONE add does the xor-32 step
// of both; v_permlane16_swap does the same for the xor-16 step of four rows.  20 VALU operations instead of 80, and the two results
// hold the totals of rows (0, 2, 1, 3) / (4, 6, 5, 7) in their four 16-lane rows -- the division and the rsqrt run once for four rows.
__device__ __forceinline__ void wave_sum8(const float (&v)[8], float& o0123, float& o4567) {
  float h[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[2 * k]), __float_as_uint(v[2 * k + 1]), false, false);
    h[k] = __uint_as_float(r[0]) + __uint_as_float(r[1]);     // lanes 0-31: row 2k, lanes 32-63: row 2k + 1
  }
  float q[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(h[2 * k]), __float_as_uint(h[2 * k + 1]), false, false);
    q[k] = __uint_as_float(r[0]) + __uint_as_float(r[1]);     // 16-lane rows: rows 4k + (0, 2, 1, 3)
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    q[k] += aql_dpp_row<0x128>(q[k]);
    q[k] += aql_dpp_row<0x124>(q[k]);
    q[k] += aql_dpp_row<0x122>(q[k]);
    q[k] += aql_dpp_row<0x121>(q[k]);
  }
  o0123 = q[0];
  o4567 = q[1];
}
// row r (0..7) of a wave_sum8 result as a wave-uniform value: 16-lane row (0, 2, 1, 3)[r & 3] of the r >> 2 register
__device__ __forceinline__ float pick8(float a0123, float a4567, int r) {
  const int lanerow = ((r & 1) << 1) | ((r >> 1) & 1);
  return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint((r & 4) ? a4567 : a0123), lanerow * 16));
}

// LayerNorm of EIGHT 320-wide rows, each held as 16-byte chunks by lanes 0..39: the arithmetic of ln_kernel<0, R, 1> (aql_norm.hip),
// operation for operation per row
__device__ __forceinline__ void ln_rows8(uint4 (&x)[8], bool act, const uint4& gr, const uint4& br, float eps, float& mean0123,
                                         float& mean4567, float& rstd0123, float& rstd4567) {
#pragma clang fp contract(off)
  float xv[8][8], s[8], vv[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const uint32_t w[4] = {x[r].x, x[r].y, x[r].z, x[r].w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      xv[r][2 * e] = bf16lo(w[e]);
      xv[r][2 * e + 1] = bf16hi(w[e]);
    }
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) t += xv[r][j];
    s[r] = act ? t : 0.f;
  }
  float sa, sb;
  wave_sum8(s, sa, sb);
  mean0123 = sa / CH;
  mean4567 = sb / CH;
  float mean[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) mean[r] = pick8(mean0123, mean4567, r);
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float d = xv[r][j] - mean[r];
      t = fmaf(d, d, t);
    }
    vv[r] = act ? t : 0.f;
  }
  wave_sum8(vv, sa, sb);
  rstd0123 = rsqrtf(sa / CH + eps);
  rstd4567 = rsqrtf(sb / CH + eps);
  float ga[8], be[8];
  const uint32_t gw[4] = {gr.x, gr.y, gr.z, gr.w}, bw[4] = {br.x, br.y, br.z, br.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    ga[2 * e] = bf16lo(gw[e]);
    ga[2 * e + 1] = bf16hi(gw[e]);
    be[2 * e] = bf16lo(bw[e]);
    be[2 * e + 1] = bf16hi(bw[e]);
  }
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const float rstd = pick8(rstd0123, rstd4567, r);
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = fmaf((xv[r][j] - mean[r]) * rstd, ga[j], be[j]);
    x[r] = make_uint4(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7]));
  }
}

typedef uint32_t u32x4_t_fwd __attribute__((ext_vector_type(4)));
// LayerNorm BACKWARD of EIGHT rows (lanes 0..39 one 16-byte chunk each): the arithmetic of ln_kernel<1, R, 1> (aql_norm.hip) per row --
//   dv = dy * gamma;  xh = (x - mean) * rstd;  s1 = sum(dv) / C;  s2 = sum(dv * xh) / C;  dx = rstd * fma(-xh, s2, dv - s1)  (+ dres)
// with the two row sums of all eight rows reduced by wave_sum8.  dy[r] is replaced by dx.
__device__ __forceinline__ void lnb_rows8(uint4 (&dy)[8], const uint4 (&x)[8], const float (&mean)[8], const float (&rstd)[8], bool act,
                                          const uint4& gr, const u32x4_t_fwd (&dres)[8], bool has_dres) {
#pragma clang fp contract(off)
  float dv[8][8], xh[8][8], s1[8], s2[8], ga[8];
  const uint32_t gw[4] = {gr.x, gr.y, gr.z, gr.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    ga[2 * e] = bf16lo(gw[e]);
    ga[2 * e + 1] = bf16hi(gw[e]);
  }
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const uint32_t dw[4] = {dy[r].x, dy[r].y, dy[r].z, dy[r].w}, xw[4] = {x[r].x, x[r].y, x[r].z, x[r].w};
    float a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int j = 2 * e + h;
        const float d0 = h ? bf16hi(dw[e]) : bf16lo(dw[e]), x0 = h ? bf16hi(xw[e]) : bf16lo(xw[e]);
        dv[r][j] = d0 * ga[j];
        xh[r][j] = (x0 - mean[r]) * rstd[r];
        a1 += dv[r][j];
        a2 = fmaf(dv[r][j], xh[r][j], a2);
      }
    }
    s1[r] = act ? a1 : 0.f;
    s2[r] = act ? a2 : 0.f;
  }
  float p0, p1, q0, q1;
  wave_sum8(s1, p0, p1);
  wave_sum8(s2, q0, q1);
  p0 = p0 / CH;
  p1 = p1 / CH;
  q0 = q0 / CH;
  q1 = q1 / CH;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const float m1 = pick8(p0, p1, r), m2 = pick8(q0, q1, r);
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = rstd[r] * fmaf(-xh[r][j], m2, dv[r][j] - m1);
    if (has_dres) {
      const uint32_t rw[4] = {dres[r].x, dres[r].y, dres[r].z, dres[r].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[2 * e] += bf16lo(rw[e]);
        o[2 * e + 1] += bf16hi(rw[e]);
      }
    }
    dy[r] = make_uint4(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7]));
  }
}

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));

// STORES: every 16-byte buffer store of this kernel takes its row offset in the VECTOR offset, never in the scalar `soffset` operand.
// Measured on MI355X (tools/stress_chain2.py, round 5): `buffer_store_dwordx4 v[a:a+3], voff, rsrc, sN offen` with an SGPR soffset reads
// its data registers LATE when the memory pipeline is backed up -- a VALU write to v[a] fourteen instructions behind the store (the
// register allocator had re-used it for a division residual) reached HBM in lanes 12-15 / 28-31 of one row in ~1 launch of 5 under
// load, never on an idle chip.  hipcc's hazard recognizer treats the SGPR-soffset form as hazard-free and inserts nothing; with the
// offset in the VGPR (soffset = 0) it applies its store-data rule and 1440 stressed launches are clean.
//
// keep a wave-uniform value in a scalar register: without this the compiler re-reads stage parameters from the kernel-argument
// segment inside the K loop and the row pass (s_load + s_waitcnt lgkmcnt(0), which also drains the LDS reads in flight)
__device__ __forceinline__ uint32_t keep_s(uint32_t v) {
  asm volatile("" : "+s"(v));
  return v;
}
__device__ __forceinline__ const bf16_t* keep_p(const bf16_t* p) {
  uint64_t v = (uint64_t)p;
  uint32_t lo_ = (uint32_t)v, hi_ = (uint32_t)(v >> 32);
  asm volatile("" : "+s"(lo_), "+s"(hi_));
  return (const bf16_t*)(((uint64_t)hi_ << 32) | lo_);
}
template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// at most 3 + n operations outstanding: n = the stores of the previous stage's DIRECT epilogue (5 FM) or row pass (8 FM rows out,
// + 2 FM statistics + 8 FM rows nout for a forward LayerNorm); anything else waits for the stores too
__device__ __forceinline__ void wait_tiles(int n) {
  switch (n) {
    case 0: wait_vm<3>(); break;
    case 5: wait_vm<8>(); break;
    case 8: wait_vm<11>(); break;
    case 10: wait_vm<13>(); break;
    case 16: wait_vm<19>(); break;
    case 18: wait_vm<21>(); break;
    case 36: wait_vm<39>(); break;
    default: wait_vm<3>(); break;
  }
}

// at most n operations outstanding (n from the small set the wide kernel's ring depths and store counts produce)
__device__ __forceinline__ void wait_le(int n) {
  switch (n) {
    case 3: wait_vm<3>(); break;
    case 8: wait_vm<8>(); break;
    case 9: wait_vm<9>(); break;
    case 11: wait_vm<11>(); break;
    case 13: wait_vm<13>(); break;
    case 14: wait_vm<14>(); break;
    case 17: wait_vm<17>(); break;
    case 19: wait_vm<19>(); break;
    case 21: wait_vm<21>(); break;
    case 27: wait_vm<27>(); break;
    default: wait_vm<3>(); break;
  }
}

// ---- row pass over the resident tile: wavefront w owns rows RW w .. RW w + RW - 1, lanes 0..39 one 16-byte chunk each (whole
// 640-byte rows to / from HBM).  Branch-free: lanes 40..63 compute on chunk 0 and are switched off by out-of-range buffer offsets /
// a dummy LDS address.  Forward: tile + residual -> out; LayerNorm -> statistics, nout.  Backward: LayerNorm backward of the tile
// (the incoming gradient) with the saved input rows and statistics, + the residual branch's gradient -> out.
// BM = rows of the tile, RW = rows per wavefront, OFF_G / OFF_B = gamma / beta images in LDS, OFF_DUMMY = 1 KB the inactive lanes may scribble on.
template <int FM, bool BWD, int OFF_G, int OFF_B, int OFF_DUMMY>
__device__ __forceinline__ int row_pass_fn(char* lds, const Stage& s, int m0, int wave, int lane) {
  constexpr int BM = 64 * FM, RW = 8 * FM;
  const int l15 = lane & 15, q4 = lane >> 4;
    const bool act = lane < CH / 8;
    const int cl = act ? lane : 0;
    const int mode = s.ln;
    const bool has_res = s.res != nullptr;
    const bool wr_n = mode == RP_LN_FWD && s.nout != nullptr && m0 >= s.nout_row0;    // block-uniform
    const float eps = s.eps;
    const uint32_t ldr2 = keep_s((uint32_t)(s.ldr * 2)), ldo2 = keep_s((uint32_t)(s.ldo * 2)), ldn2 = keep_s((uint32_t)(s.ldn * 2)),
                   ldx2 = keep_s((uint32_t)(s.ldlx * 2));
    const __amdgpu_buffer_rsrc_t rsR = make_rsrc(s.res), rsO = make_rsrc(s.out), rsN = make_rsrc(wr_n ? s.nout : nullptr),
                                 rsS = make_rsrc(mode != RP_NONE ? s.stats : nullptr), rsX = make_rsrc(mode == RP_LN_BWD ? s.lnx : nullptr);
    const uint32_t mrow = (uint32_t)(m0 + wave * RW);
    const uint32_t vr = mrow * ldr2 + cl * 16, vx = mrow * ldx2 + cl * 16;
    const uint32_t vo = act ? mrow * ldo2 + cl * 16 : OOB_ROW;
    const uint32_t vn = act ? mrow * ldn2 + cl * 16 : OOB_ROW;
    const int myrow = ((q4 & 1) << 1) | (q4 >> 1);             // the row (of four) whose statistics this lane's 16-lane row holds
    const uint32_t vs4 = l15 == 0 ? (mrow + myrow) * 8u : OOB_ROW;
    const int kbase = (cl >> 2) * (BM * 64);
    auto laddr = [&](int rr) __attribute__((always_inline)) {
      const int row = wave * RW + rr;      // wave-uniform
      return kbase + row * 64 + (((cl & 3) ^ swz4(row)) << 4);
    };
    uint4 gr = make_uint4(0u, 0u, 0u, 0u), br = gr;
    if (mode != RP_NONE) {
      gr = *reinterpret_cast<const uint4*>(lds + OFF_G + cl * 16);
      br = *reinterpret_cast<const uint4*>(lds + OFF_B + cl * 16);
    }
    auto batch = [&](auto res_tag, auto mode_tag, int r0) __attribute__((always_inline)) {
      constexpr bool RES = decltype(res_tag)::value;
      constexpr int MODE = decltype(mode_tag)::value;
      uint4 xv[8];
      u32x4_t rv[8];
      if constexpr (MODE == RP_LN_BWD) {
        u32x4_t xs[8];
        u32x2_t st[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int rr = r0 + u;
          xs[u] = __builtin_amdgcn_raw_buffer_load_b128(rsX, vx + rr * ldx2, 0, 0);
          st[u] = __builtin_amdgcn_raw_buffer_load_b64(rsS, (mrow + rr) * 8u, 0, 0);       // the row's (mean, rstd): one address for the wavefront
          if constexpr (RES) rv[u] = __builtin_amdgcn_raw_buffer_load_b128(rsR, vr + rr * ldr2, 0, 0);
          xv[u] = *reinterpret_cast<const uint4*>(lds + laddr(rr));
        }
        uint4 xin[8];
        float mean[8], rstd[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          xin[u] = make_uint4(xs[u].x, xs[u].y, xs[u].z, xs[u].w);
          mean[u] = __uint_as_float(st[u].x);
          rstd[u] = __uint_as_float(st[u].y);
        }
        lnb_rows8(xv, xin, mean, rstd, act, gr, rv, RES);
#pragma unroll
        for (int u = 0; u < 8; ++u)
          __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{xv[u].x, xv[u].y, xv[u].z, xv[u].w}, rsO, vo + (r0 + u) * ldo2, 0, 0);
      } else {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int rr = r0 + u;
          if constexpr (RES) rv[u] = __builtin_amdgcn_raw_buffer_load_b128(rsR, vr + rr * ldr2, 0, 0);
          xv[u] = *reinterpret_cast<const uint4*>(lds + laddr(rr));
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int rr = r0 + u;
          if constexpr (RES) xv[u] = epi_add8(xv[u], make_uint4(rv[u].x, rv[u].y, rv[u].z, rv[u].w));
          __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{xv[u].x, xv[u].y, xv[u].z, xv[u].w}, rsO, vo + rr * ldo2, 0, 0);   // null descriptor: dropped
        }
        if constexpr (MODE == RP_LN_FWD) {
          float ma, mb, ra, rb;
          ln_rows8(xv, act, gr, br, eps, ma, mb, ra, rb);
          __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{__float_as_uint(ma), __float_as_uint(ra)}, rsS, vs4 + r0 * 8, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{__float_as_uint(mb), __float_as_uint(rb)}, rsS, vs4 + (r0 + 4) * 8, 0, 0);
#pragma unroll
          for (int u = 0; u < 8; ++u)
            __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{xv[u].x, xv[u].y, xv[u].z, xv[u].w}, rsN, vn + (r0 + u) * ldn2, 0, 0);
        }
      }
      if constexpr (RES || MODE != RP_NONE) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          char* const wp = act ? lds + laddr(r0 + u) : lds + OFF_DUMMY + lane * 16;
          *reinterpret_cast<uint4*>(wp) = xv[u];
        }
      }
    };
    auto pass = [&](auto res_tag, auto mode_tag) __attribute__((always_inline)) {
#pragma unroll
      for (int b = 0; b < FM; ++b) batch(res_tag, mode_tag, 8 * b);
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    if constexpr (BWD) {
      if (has_res) pass(T_{}, std::integral_constant<int, RP_LN_BWD>{});
      else pass(F_{}, std::integral_constant<int, RP_LN_BWD>{});
    } else {
      if (mode == RP_LN_FWD) {
        if (has_res) pass(T_{}, std::integral_constant<int, RP_LN_FWD>{});
        else pass(F_{}, std::integral_constant<int, RP_LN_FWD>{});
      } else {
        if (has_res) pass(T_{}, std::integral_constant<int, RP_NONE>{});
        else pass(F_{}, std::integral_constant<int, RP_NONE>{});
      }
    }
  return mode == RP_LN_FWD ? 18 * FM : 8 * FM;   // stores ISSUED per wavefront (dropped ones included)
}

// BWD: the row passes of this instance are LayerNorm BACKWARD passes (the backward chains); else forward (residual / LayerNorm).  Two
// instances instead of a run-time mode: the backward pass holds 128 more floats per lane across its reductions, and one kernel with
// both would take its register allocation (spills in the forward K loop).
template <int FM, bool BWD>
__global__ __launch_bounds__(NTH, 2) void chain_kernel(const Args a) {
  using LY = Lay<FM>;
  constexpr int BM = LY::BM;                  // rows of the tile
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  // tile of this workgroup; twin batch: clean and LoRA row tiles alternate over the block index
  const int tiles_m = a.M / BM;
  int tile_m = blockIdx.x;
  if (a.row0 > 0 && (tiles_m & 1) == 0 && a.row0 == (tiles_m >> 1) * BM)
    tile_m = (tile_m & 1) ? (tiles_m >> 1) + (tile_m >> 1) : (tile_m >> 1);
  const int m0 = tile_m * BM;
  const bool lora_tile = m0 + BM > a.row0;   // block-uniform
  int mark_ = 0;
  long long* const trc = (a.trace != nullptr && lane == 0 && (wave == 0 || wave == 4)) ? a.trace + ((long)blockIdx.x * 2 + (wave >> 2)) * 32 : nullptr;
#ifdef AQL_CHAIN_TRACE   // tools/trace_chain.py on a trace build (tools/build_alt.sh trace aql_chain.hip -DAQL_CHAIN_TRACE=1)
#define CH_STAMP() do { if (trc != nullptr && mark_ < 32) trc[mark_] = __builtin_readcyclecounter(); ++mark_; } while (0)
#else
#define CH_STAMP() do { (void)trc; (void)mark_; } while (0)
#endif
  CH_STAMP();   // 0 start

  // ---- lane constants
  const int l15 = lane & 15, q4 = lane >> 4;
  const int lo = l15 * 64 + ((q4 ^ swz4(l15)) << 4);                    // fragment read: row base16 + l15, chunk q4
  // DMA: a wave instruction fills 16 rows x 64 B; lane -> (row lane >> 2, physical slot lane & 3) fetches logical chunk slot ^ swz
  const int drow = lane >> 2;
  const uint32_t dchunk = (uint32_t)(((lane & 3) ^ swz4(drow)) << 4);
  const uint32_t dr0 = (uint32_t)(wave * 16 + drow);

  // ---- constants of the chain into LDS: biases, gamma / beta of the (single) LayerNorm, the tile's scale row
  for (int id = tid; id < a.nstage * (CH / 8); id += NTH) {
    const int g = id / (CH / 8), c = id - g * (CH / 8);
    const bf16_t* b = a.st[g].bias;
    const uint4 v = b ? *reinterpret_cast<const uint4*>(b + c * 8) : make_uint4(0u, 0u, 0u, 0u);
    *reinterpret_cast<uint4*>(lds + LY::OFF_BIAS + g * CH * 2 + c * 16) = v;
  }
  if (tid < CH / 8) {
    const bf16_t *gm = a.has_pre ? a.pre.gamma : nullptr, *bt = a.has_pre ? a.pre.beta : nullptr;
    for (int g = 0; g < a.nstage; ++g)
      if (a.st[g].ln) gm = a.st[g].gamma, bt = a.st[g].beta;
    if (gm) *reinterpret_cast<uint4*>(lds + LY::OFF_GAMMA + tid * 16) = *reinterpret_cast<const uint4*>(gm + tid * 8);
    if (bt) *reinterpret_cast<uint4*>(lds + LY::OFF_BETA + tid * 16) = *reinterpret_cast<const uint4*>(bt + tid * 8);
  }
  if (tid < 4)
    *reinterpret_cast<uint4*>(lds + LY::OFF_SROW + tid * 16) =
        (lora_tile && a.S) ? *reinterpret_cast<const uint4*>(a.S + (long)(m0 / a.rps) * LR + tid * 8) : make_uint4(0u, 0u, 0u, 0u);

  // ---- the chain input tile -> resident region: 10 K tiles x BM / 16 instructions of 1 KB, 5 FM per wavefront
  {
    const __amdgpu_buffer_rsrc_t rsX = make_rsrc(a.X);
#pragma unroll
    for (int i = 0; i < 5 * FM; ++i) {
      const int q = wave + 8 * i, kt = q / (BM / 16), rb = q % (BM / 16);
      const uint32_t voff = (uint32_t)(m0 + rb * 16 + drow) * (uint32_t)(a.ldx * 2) + dchunk;
      dma16(rsX, lds + kt * (BM * 64) + rb * 1024, voff, (uint32_t)kt * 64u);
    }
  }

  // ---- weight-tile stream.  Three LDS-DMA instructions per wavefront and tile: W rows 16 w.., 128 + 16 w.., and either W rows
  // 256 + 16 w.. (wavefronts 0-3) or 16 rows of the LoRA down tile (wavefronts 4-7; 6 / 7 duplicate 4 / 5 -- the count per
  // wavefront is what the vmcnt waits rely on).  Tiles are requested two ahead of their use.
  int wr = 0, rd = 0;
  auto ring_next = [](int x) { return x + 1 == NSTG ? 0 : x + 1; };
  // W / LoRA-down K tile kt of one linear
  auto issue_w = [&](const __amdgpu_buffer_rsrc_t& rsW, uint32_t ldb, const __amdgpu_buffer_rsrc_t& rsL, int kt) __attribute__((always_inline)) {
    char* dst = lds + LY::OFF_RING + wr * STAGE;
    const uint32_t soff = (uint32_t)kt * 64u, v0 = dr0 * ldb + dchunk;
    dma16(rsW, dst + wave * 1024, v0, soff);
    dma16(rsW, dst + (wave + 8) * 1024, v0 + 128u * ldb, soff);
    if (wave < 4) dma16(rsW, dst + (wave + 16) * 1024, v0 + 256u * ldb, soff);
    else dma16(rsL, dst + W_BYTES + (wave & 1) * 1024, (uint32_t)((wave & 1) * 16 + drow) * (uint32_t)(CH * 2) + dchunk, soff);   // a null descriptor zero-fills
    wr = ring_next(wr);
  };
  auto issue_up = [&](const __amdgpu_buffer_rsrc_t& rsB) __attribute__((always_inline)) {   // the Bup tile [320][32]
    char* dst = lds + LY::OFF_RING + wr * STAGE;
    const uint32_t v0 = dr0 * 64u + dchunk;
    dma16(rsB, dst + wave * 1024, v0, 0);
    dma16(rsB, dst + (wave + 8) * 1024, v0 + 128u * 64u, 0);
    if (wave < 4) dma16(rsB, dst + (wave + 16) * 1024, v0 + 256u * 64u, 0);
    else dma16(make_rsrc(nullptr), dst + W_BYTES + (wave & 1) * 1024, OOB_ROW, 0);
    wr = ring_next(wr);
  };
  {
    const Stage& s0 = a.st[0];
    const __amdgpu_buffer_rsrc_t rsW = make_rsrc(s0.W), rsL = make_rsrc((lora_tile && s0.Ad) ? s0.Ad : nullptr);
    issue_w(rsW, (uint32_t)(s0.ldw * 2), rsL, 0);
    issue_w(rsW, (uint32_t)(s0.ldw * 2), rsL, 1);
  }

  const int aoff = (wm * 16 * FM) * 64 + lo;       // A fragments: rows wm * 16 FM + 16 i + l15 of a K tile
  const int boff = (wn * 160) * 64 + lo;           // W fragments: rows wn*160 + 16 j + l15
  const int loff = W_BYTES + (wn * 16) * 64 + lo;  // LoRA-down fragment: rank rows wn*16 + l15
  int pend = 0;                                    // stores of the previous stage still counted by vmcnt (per wavefront)

  auto row_pass = [&](const Stage& s) __attribute__((always_inline)) {
    return row_pass_fn<FM, BWD, LY::OFF_GAMMA, LY::OFF_BETA, LY::OFF_TS>(lds, s, m0, wave, lane);
  };

  if (a.has_pre) {   // backward chains: the incoming gradient goes through the LayerNorm backward before the first linear
    wait_vm<6>();    // the input tile has landed (the two weight tiles behind it may still fly)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    pend = row_pass(a.pre);
  }

  for (int g = 0; g < a.nstage; ++g) {
    const Stage& s = a.st[g];
    const bool lora_g = lora_tile && s.Ad != nullptr;
    const bool has_next = g + 1 < a.nstage;
    // stage parameters into scalar registers, once
    const bf16_t* const pW = keep_p(s.W);
    const bf16_t* const pAd = keep_p(lora_g ? s.Ad : nullptr);
    const bf16_t* const pBup = keep_p(lora_g ? s.Bup : nullptr);
    const uint32_t ldb = keep_s((uint32_t)(s.ldw * 2));
    const Stage& sn = a.st[has_next ? g + 1 : g];
    const bf16_t* const pWn = keep_p(has_next ? sn.W : nullptr);
    const bf16_t* const pAdn = keep_p((has_next && lora_tile && sn.Ad) ? sn.Ad : nullptr);
    const uint32_t ldbn = keep_s((uint32_t)(sn.ldw * 2));
    const int keep = (int)keep_s((uint32_t)s.keep);
    const __amdgpu_buffer_rsrc_t rsW = make_rsrc(pW), rsL = make_rsrc(pAd);
    // tile t + 2 of this stage's stream, requested in iteration t: W tiles 2..9, then the Bup tile (LoRA), then the next stage's
    // tiles 0 / 1 (or zero fills past the end of the chain: a null descriptor)
    auto issue_ahead = [&](int t2) __attribute__((always_inline)) {
      if (t2 < NKT) {
        issue_w(rsW, ldb, rsL, t2);
      } else if (lora_g && t2 == NKT) {
        issue_up(make_rsrc(pBup));
      } else {
        issue_w(make_rsrc(pWn), ldbn, make_rsrc(pAdn), t2 - NKT - (lora_g ? 1 : 0));
      }
    };

    f32x4_t acc[FM][10], tacc[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
#pragma unroll
      for (int j = 0; j < 10; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      tacc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    // two copies of the K loop (with / without the T side product), chosen once per stage: a uniform branch inside the k-step cuts
    // the compiler's ds_read / MFMA pipeline (aql_gemm_lora.hip)
    auto mainloop = [&](auto lora_tag) __attribute__((always_inline)) {
      constexpr bool LORA = decltype(lora_tag)::value;
#pragma unroll
      for (int t = 0; t < NKT; ++t) {
        // iterations 0 / 1: the previous stage's stores are YOUNGER than this tile's requests -- count past them; from
        // iteration 2 on the awaited tile is younger than the stores, which have had two tiles' time to drain
        if (t < 2) wait_tiles(pend);
        else wait_vm<3>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t == 0) CH_STAMP();   // 1 + 6 g: the stage's first K tile (g = 0: and the chain input) has landed
        const char* sW = lds + LY::OFF_RING + rd * STAGE;
        const char* sA = lds + t * (BM * 64);
        bf16x8_t fa[FM], fb[10], fl;
#pragma unroll
        for (int i = 0; i < FM; ++i) fa[i] = *reinterpret_cast<const bf16x8_t*>(sA + aoff + i * 1024);
#pragma unroll
        for (int j = 0; j < 10; ++j) fb[j] = *reinterpret_cast<const bf16x8_t*>(sW + boff + j * 1024);
        if constexpr (LORA) fl = *reinterpret_cast<const bf16x8_t*>(sW + loff);
#pragma unroll
        for (int j = 0; j < 10; ++j) acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[0], acc[0][j], 0, 0, 0);
        if constexpr (LORA) tacc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fl, fa[0], tacc[0], 0, 0, 0);
        issue_ahead(t + 2);
        if constexpr (FM == 2) {
#pragma unroll
          for (int j = 0; j < 10; ++j) acc[FM - 1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[FM - 1], acc[FM - 1][j], 0, 0, 0);
          if constexpr (LORA) tacc[FM - 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fl, fa[FM - 1], tacc[FM - 1], 0, 0, 0);
        }
        rd = ring_next(rd);
      }
    };
    if (lora_g) mainloop(std::true_type{});
    else mainloop(std::false_type{});
    pend = 0;
    CH_STAMP();   // 2 + 6 g: K loop issued

    if (lora_g) {
      // ---- T -> (T, Ts) bf16; Ts as one more A tile; one k-step against the Bup tile of the ring
      const uint2 sv = *reinterpret_cast<const uint2*>(lds + LY::OFF_SROW + (wn * 16 + q4 * 4) * 2);
      const __amdgpu_buffer_rsrc_t rsT = make_rsrc(s.T), rsTs = make_rsrc(s.Ts);
      const uint32_t vt = (uint32_t)(m0 + wm * 16 * FM + l15) * 64u + (uint32_t)(wn * 16 + q4 * 4) * 2u;
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int row = wm * 16 * FM + i * 16 + l15;
        const int r = wn * 16 + q4 * 4;
        const u32x2_t tv = {pack_bf16x2(tacc[i][0], tacc[i][1]), pack_bf16x2(tacc[i][2], tacc[i][3])};
        const u32x2_t ts = {pack_bf16x2(bf16lo(tv.x) * bf16lo(sv.x), bf16hi(tv.x) * bf16hi(sv.x)),
                            pack_bf16x2(bf16lo(tv.y) * bf16lo(sv.y), bf16hi(tv.y) * bf16hi(sv.y))};
        *reinterpret_cast<u32x2_t*>(lds + LY::OFF_TS + off64(row, r >> 3) + (r & 7) * 2) = ts;
        __builtin_amdgcn_raw_buffer_store_b64(tv, rsT, vt + i * 1024, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b64(ts, rsTs, vt + i * 1024, 0, 0);
      }
      wait_vm<3 + 3 + 2 * FM>();   // in order: [Bup tile 3] [next tile 3] [T / Ts stores 2 FM]
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const char* sW = lds + LY::OFF_RING + rd * STAGE;
      bf16x8_t fa[FM], fb[10];
#pragma unroll
      for (int i = 0; i < FM; ++i) fa[i] = *reinterpret_cast<const bf16x8_t*>(lds + LY::OFF_TS + aoff + i * 1024);
#pragma unroll
      for (int j = 0; j < 10; ++j) fb[j] = *reinterpret_cast<const bf16x8_t*>(sW + boff + j * 1024);
#pragma unroll
      for (int j = 0; j < 10; ++j) acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[0], acc[0][j], 0, 0, 0);
      issue_ahead(NKT + 2);
      if constexpr (FM == 2) {
#pragma unroll
        for (int j = 0; j < 10; ++j) acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[1], acc[1][j], 0, 0, 0);
      }
      rd = ring_next(rd);
    } else if (keep) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();   // every wavefront has read its last resident K tile: the epilogue may overwrite it
      asm volatile("" ::: "memory");
    }
    CH_STAMP();   // 3 + 6 g: up step issued

    // ---- epilogue: bias in fp32, round to bf16; column blocks pairwise through v_permlane16_swap so that a lane holds 8 consecutive
    // columns (16 bytes): lane row q4 = 0 / 2 -> block 2jp, columns 0-7 / 8-15; q4 = 1 / 3 -> block 2jp + 1
    {
      const char* sBias = lds + LY::OFF_BIAS + g * CH * 2 + (wn * 160 + q4 * 4) * 2;
      const float os = keep ? 1.f : s.oscale;                        // (x * 1.0f is exact: every other stage keeps its bits)
      const int c0 = wn * 160 + (q4 & 1) * 16 + (q4 >> 1) * 8;       // + 32 jp
      const __amdgpu_buffer_rsrc_t rsO = make_rsrc(keep ? nullptr : s.out);
      const uint32_t ldo2 = keep ? 0u : (uint32_t)(s.ldo * 2);
      const uint32_t vo = (uint32_t)(m0 + wm * 16 * FM + l15) * ldo2 + (uint32_t)c0 * 2u;
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int row = wm * 16 * FM + i * 16 + l15;
        char* const ldst = lds + row * 64;
        const int sz = swz4(row);
#pragma unroll
        for (int jp = 0; jp < 5; ++jp) {
          const uint2 bA = *reinterpret_cast<const uint2*>(sBias + jp * 64), bB = *reinterpret_cast<const uint2*>(sBias + jp * 64 + 32);
          const f32x4_t& xa = acc[i][2 * jp];
          const f32x4_t& xb = acc[i][2 * jp + 1];
          uint32_t x0 = pack_bf16x2((xa[0] + bf16lo(bA.x)) * os, (xa[1] + bf16hi(bA.x)) * os), x1 = pack_bf16x2((xa[2] + bf16lo(bA.y)) * os, (xa[3] + bf16hi(bA.y)) * os);
          uint32_t y0 = pack_bf16x2((xb[0] + bf16lo(bB.x)) * os, (xb[1] + bf16hi(bB.x)) * os), y1 = pack_bf16x2((xb[2] + bf16lo(bB.y)) * os, (xb[3] + bf16hi(bB.y)) * os);
          const auto s0 = __builtin_amdgcn_permlane16_swap(x0, y0, false, false);
          const auto s1 = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
          const u32x4_t v = {s0[0], s1[0], s0[1], s1[1]};
          const int cc = c0 + 32 * jp;
          if (keep) *reinterpret_cast<u32x4_t*>(ldst + (cc >> 5) * (BM * 64) + ((((cc >> 3) & 3) ^ sz) << 4)) = v;
          else __builtin_amdgcn_raw_buffer_store_b128(v, rsO, vo + i * 16 * ldo2 + jp * 64, 0, 0);
        }
      }
    }
    CH_STAMP();   // 4 + 6 g: epilogue done
    if (!keep) {
      CH_STAMP();
      CH_STAMP();
      pend = 5 * FM;   // FM x 5 output stores per wavefront
      continue;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    CH_STAMP();   // 5 + 6 g: tile published
    pend = row_pass(s);
    CH_STAMP();   // 6 + 6 g: row pass done
    // the next stage's first barrier publishes the rewritten tile
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // trailing zero-fill DMAs still target the LDS
  CH_STAMP();   // everything drained
#undef CH_STAMP
}

// ------------------------------------------------------------------------------------------------------------------------------------
// RANK-320 chains (BASELINE config 3: train/README.md:34-48 trains at rank 320).  The rank-32 kernel lets the down product ride in the K
// loop; at rank 320 the down product T = X.A^T is a 320 x 320 GEMM of its own and Ts = T * S is a second FULL-WIDTH A panel.  With 64-row
// tiles both panels fit the LDS next to the weight ring (40 + 40 + 60 KB), so the whole LoRA linear runs inside the chain:
//     pass T:    acc = X.A^T                      -> T (bf16), Ts = T * S[sample]  -> HBM (backward) and the Ts panel in LDS
//     pass main: acc = X.W^T + Ts.Bup^T + bias    -> the epilogues of the rank-32 kernel (DIRECT / KEEP + row pass)
// i.e. 30 weight tiles of 320 x 32 per stage instead of 11 (10 on tiles of the clean half).  Same roundings in the same places as
// aql_lora_down (T, Ts through the row-scaled second output) + the two-K-segment aql_gemm_bf16: bit-identical (tools/probe_chain.py).
struct LayW {
  static constexpr int BM = 64;
  static constexpr int OFF_TSW = NKT * BM * 64;             // 40960: the Ts panel, 10 K tiles x 64 rows x 64 B
  static constexpr int OFF_RING = 2 * OFF_TSW;              // 3 stages of 320 x 64 B
  static constexpr int OFF_DUMMY = OFF_RING + NSTG * W_BYTES;   // 2 KB nobody reads: wavefronts 4-7's third DMA, inactive lanes of the row pass
  static constexpr int OFF_BIAS = OFF_DUMMY + 2048;
  static constexpr int OFF_GAMMA = OFF_BIAS + MAXS * CH * 2;
  static constexpr int OFF_BETA = OFF_GAMMA + CH * 2;
  static constexpr int OFF_SROW = OFF_BETA + CH * 2;        // the tile's scale row: 320 bf16
  static constexpr int TOTAL = OFF_SROW + CH * 2;
};
static_assert(LayW::TOTAL <= 160 * 1024, "LDS budget");

__global__ __launch_bounds__(NTH, 2) void chain_wide_kernel(const Args a) {
  using LY = LayW;
  constexpr int BM = LY::BM;
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;     // wavefront grid 2 (M) x 4 (N): 32 rows x 80 columns each -- 7 fragment reads per 10 MFMAs
  const int tiles_m = a.M / BM;
  int tile_m = blockIdx.x;
  if (a.row0 > 0 && a.row0 < a.M) {            // twin batch: the LoRA tiles (3x the work) first, the clean tiles fill in behind them
    const int t0 = a.row0 / BM;
    tile_m = tile_m < tiles_m - t0 ? t0 + tile_m : tile_m - (tiles_m - t0);
  }
  const int m0 = tile_m * BM;
  const bool lora_tile = m0 + BM > a.row0;   // block-uniform
  int mark_ = 0;
  long long* const trc = (a.trace != nullptr && lane == 0 && (wave == 0 || wave == 4)) ? a.trace + ((long)blockIdx.x * 2 + (wave >> 2)) * 32 : nullptr;
#ifdef AQL_CHAIN_TRACE
#define CW_STAMP() do { if (trc != nullptr && mark_ < 32) trc[mark_] = __builtin_readcyclecounter(); ++mark_; } while (0)
#else
#define CW_STAMP() do { (void)trc; (void)mark_; } while (0)
#endif
  CW_STAMP();   // 0 start (tools/trace_chain.py wide)
  const int l15 = lane & 15, q4 = lane >> 4;
  const int lo = l15 * 64 + ((q4 ^ swz4(l15)) << 4);
  const int drow = lane >> 2;
  const uint32_t dchunk = (uint32_t)(((lane & 3) ^ swz4(drow)) << 4);
  const uint32_t dr0 = (uint32_t)(wave * 16 + drow);

  for (int id = tid; id < a.nstage * (CH / 8); id += NTH) {
    const int g = id / (CH / 8), c = id - g * (CH / 8);
    const bf16_t* b = a.st[g].bias;
    const uint4 v = b ? *reinterpret_cast<const uint4*>(b + c * 8) : make_uint4(0u, 0u, 0u, 0u);
    *reinterpret_cast<uint4*>(lds + LY::OFF_BIAS + g * CH * 2 + c * 16) = v;
  }
  if (tid < CH / 8) {
    const bf16_t *gm = nullptr, *bt = nullptr;
    for (int g = 0; g < a.nstage; ++g)
      if (a.st[g].ln) gm = a.st[g].gamma, bt = a.st[g].beta;
    if (gm) *reinterpret_cast<uint4*>(lds + LY::OFF_GAMMA + tid * 16) = *reinterpret_cast<const uint4*>(gm + tid * 8);
    if (bt) *reinterpret_cast<uint4*>(lds + LY::OFF_BETA + tid * 16) = *reinterpret_cast<const uint4*>(bt + tid * 8);
    *reinterpret_cast<uint4*>(lds + LY::OFF_SROW + tid * 16) =
        (lora_tile && a.S) ? *reinterpret_cast<const uint4*>(a.S + (long)(m0 / a.rps) * CH + tid * 8) : make_uint4(0u, 0u, 0u, 0u);
  }
  {
    const __amdgpu_buffer_rsrc_t rsX = make_rsrc(a.X);
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int q = wave + 8 * i, kt = q / (BM / 16), rb = q % (BM / 16);
      const uint32_t voff = (uint32_t)(m0 + rb * 16 + drow) * (uint32_t)(a.ldx * 2) + dchunk;
      dma16(rsX, lds + kt * (BM * 64) + rb * 1024, voff, (uint32_t)kt * 64u);
    }
  }

  auto tile_body = [&](auto ns_tag) __attribute__((always_inline)) {
    // ---- weight-tile stream: segments of 10 tiles [320 rows][32 k]; a stage with LoRA on a LoRA tile has three (A_down, W, Bup), else
    // one (W).  The ISSUE cursor runs two tiles ahead of the consuming loops, across segments and stages.
    // Ring: 3 slots behind the two panels on a LoRA tile; a clean tile never builds a Ts panel and takes its 40 KB as two more slots
    // (5 slots, 4 tiles in flight -- measured neutral: the launch is made of its HBM traffic and of epilogue / row-pass phases that do
    // not overlap the K loops, profiles/r05_chain_r320.txt; kept because it is free)
    constexpr int NS = decltype(ns_tag)::value, PD = NS - 1;
    constexpr int RING0 = NS == NSTG ? LY::OFF_RING : LY::OFF_TSW;
    int wr = 0, rd = 0;
    auto ring_next = [](int x) { return x + 1 == NS ? 0 : x + 1; };
    int ig = 0, ij = 0, it_ = 0;                 // issue cursor: stage, segment, tile
    const bf16_t* ip = nullptr;                  // its segment's matrix and row pitch in bytes
    uint32_t ildb = 0;
    auto nseg_of = [&](int g) { return (lora_tile && a.st[g].Ad != nullptr) ? 3 : 1; };
    auto seg_load = [&]() __attribute__((always_inline)) {
      if (ig < a.nstage) {
        const Stage& s = a.st[ig];
        const bool wide = lora_tile && s.Ad != nullptr;
        const int kind = wide ? ij : 1;          // 0 = A_down, 1 = W, 2 = Bup
        ip = keep_p(kind == 0 ? s.Ad : kind == 1 ? s.W : s.Bup);
        ildb = keep_s(kind == 1 ? (uint32_t)(s.ldw * 2) : (uint32_t)(CH * 2));
      } else {
        ip = nullptr;
        ildb = 0;
      }
    };
    auto issue = [&]() __attribute__((always_inline)) {
      char* dst = lds + RING0 + wr * W_BYTES;
      const __amdgpu_buffer_rsrc_t rs = make_rsrc(ip);      // past the end of the chain: a null descriptor zero-fills without traffic
      const uint32_t soff = (uint32_t)it_ * 64u, v0 = dr0 * ildb + dchunk;
      dma16(rs, dst + wave * 1024, v0, soff);
      dma16(rs, dst + (wave + 8) * 1024, v0 + 128u * ildb, soff);
      if (wave < 4) dma16(rs, dst + (wave + 16) * 1024, v0 + 256u * ildb, soff);
      else dma16(make_rsrc(nullptr), lds + LY::OFF_DUMMY + (wave & 1) * 1024, OOB_ROW, 0);    // keeps the per-wavefront count at 3
      wr = ring_next(wr);
      if (++it_ == NKT) {
        it_ = 0;
        if (ig < a.nstage && ++ij == nseg_of(ig)) {
          ij = 0;
          ++ig;
        }
        seg_load();
      }
    };
    seg_load();
  #pragma unroll
    for (int i = 0; i < PD; ++i) issue();

    const int aoff = (wm * 32) * 64 + lo;
    const int boff = (wn * 80) * 64 + lo;
    int pend = 0;
    f32x4_t acc[2][5];
    auto zero_acc = [&]() __attribute__((always_inline)) {
  #pragma unroll
      for (int i = 0; i < 2; ++i)
  #pragma unroll
        for (int j = 0; j < 5; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    };
    // one segment: acc += A_panel . B^T over 10 K tiles; A_panel = the resident tile or the Ts panel
    auto segment = [&](int a_base) __attribute__((always_inline)) {
  #pragma unroll
      for (int t = 0; t < NKT; ++t) {
        if (t < PD) wait_le(3 * (PD - 1) + pend);      // the previous stage's stores are younger than the tiles in flight: count past them
        else wait_vm<3 * (PD - 1)>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const char* sW = lds + RING0 + rd * W_BYTES;
        const bf16x8_t fa0 = *reinterpret_cast<const bf16x8_t*>(lds + a_base + t * (BM * 64) + aoff);
        const bf16x8_t fa1 = *reinterpret_cast<const bf16x8_t*>(lds + a_base + t * (BM * 64) + aoff + 1024);
        bf16x8_t fb[5];
  #pragma unroll
        for (int j = 0; j < 5; ++j) fb[j] = *reinterpret_cast<const bf16x8_t*>(sW + boff + j * 1024);
  #pragma unroll
        for (int j = 0; j < 5; ++j) acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa0, acc[0][j], 0, 0, 0);
        issue();
  #pragma unroll
        for (int j = 0; j < 5; ++j) acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa1, acc[1][j], 0, 0, 0);
        rd = ring_next(rd);
      }
      pend = 0;
    };
    // accumulators (+ bias image or nothing) -> bf16; the two ROW fragments of a column block are paired through v_permlane16_swap so
    // that a lane holds 8 consecutive columns (16 bytes) of one row: lane rows q4 = 0 / 2 -> row fragment 0, columns 0-7 / 8-15 of the
    // block, q4 = 1 / 3 -> row fragment 1.  `sink(cc, v)` gets the chunk of columns cc .. cc + 7 of row `row`
    const int row = wm * 32 + (q4 & 1) * 16 + l15;
    const int c0 = wn * 80 + (q4 >> 1) * 8;      // + 16 j
    auto emit = [&](const char* sBias, const float os, auto&& sink) __attribute__((always_inline)) {
  #pragma unroll
      for (int j = 0; j < 5; ++j) {
        uint2 bA = make_uint2(0u, 0u);
        if (sBias != nullptr) bA = *reinterpret_cast<const uint2*>(sBias + j * 32);
        const f32x4_t& xa = acc[0][j];
        const f32x4_t& xb = acc[1][j];
        uint32_t x0 = pack_bf16x2((xa[0] + bf16lo(bA.x)) * os, (xa[1] + bf16hi(bA.x)) * os), x1 = pack_bf16x2((xa[2] + bf16lo(bA.y)) * os, (xa[3] + bf16hi(bA.y)) * os);
        uint32_t y0 = pack_bf16x2((xb[0] + bf16lo(bA.x)) * os, (xb[1] + bf16hi(bA.x)) * os), y1 = pack_bf16x2((xb[2] + bf16lo(bA.y)) * os, (xb[3] + bf16hi(bA.y)) * os);
        const auto s0 = __builtin_amdgcn_permlane16_swap(x0, y0, false, false);
        const auto s1 = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
        sink(c0 + 16 * j, u32x4_t{s0[0], s1[0], s0[1], s1[1]});
      }
    };
    const int sz = swz4(row);
    auto lds_chunk = [&](int base, int cc) __attribute__((always_inline)) {
      return lds + base + (cc >> 5) * (BM * 64) + row * 64 + ((((cc >> 3) & 3) ^ sz) << 4);
    };

    for (int g = 0; g < a.nstage; ++g) {
      const Stage& s = a.st[g];
      const bool wide = lora_tile && s.Ad != nullptr;
      const int keep = (int)keep_s((uint32_t)s.keep);
      if (wide) {
        // ---- pass T: T = X.A^T, Ts = T * S -> HBM and the Ts panel
        zero_acc();
        segment(0);
        CW_STAMP();   // 1 + 6 g: pass T issued
        const __amdgpu_buffer_rsrc_t rsT = make_rsrc(s.T), rsTs = make_rsrc(s.Ts);
        const uint32_t vt = (uint32_t)(m0 + row) * (uint32_t)(CH * 2);
        emit(nullptr, 1.f, [&](int cc, const u32x4_t& tv) __attribute__((always_inline)) {
          const uint4 sv = *reinterpret_cast<const uint4*>(lds + LY::OFF_SROW + cc * 2);
          const uint4 ts = epi_mul8(make_uint4(tv.x, tv.y, tv.z, tv.w), sv);
          *reinterpret_cast<uint4*>(lds_chunk(LY::OFF_TSW, cc)) = ts;
          __builtin_amdgcn_raw_buffer_store_b128(tv, rsT, vt + cc * 2, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{ts.x, ts.y, ts.z, ts.w}, rsTs, vt + cc * 2, 0, 0);
        });
        pend = 10;    // T and Ts: 2 x 5 stores, younger than the two tiles in flight
        CW_STAMP();   // 2 + 6 g: T / Ts out
      } else {
        CW_STAMP();
        CW_STAMP();
      }
      // ---- pass main: X.W^T (+ Ts.Bup^T); its first barrier publishes the Ts panel
      zero_acc();
      segment(0);
      CW_STAMP();   // 3 + 6 g: X.W^T issued
      if (wide) segment(LY::OFF_TSW);
      CW_STAMP();   // 4 + 6 g: Ts.Bup^T issued
      if (keep) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // every wavefront has read its last resident K tile: the epilogue may overwrite it
        asm volatile("" ::: "memory");
      }
      const char* sBias = lds + LY::OFF_BIAS + g * CH * 2 + (wn * 80 + q4 * 4) * 2;
      if (keep) {
        emit(sBias, 1.f, [&](int cc, const u32x4_t& v) __attribute__((always_inline)) { *reinterpret_cast<u32x4_t*>(lds_chunk(0, cc)) = v; });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        CW_STAMP();   // 5 + 6 g: epilogue, tile published
        pend = row_pass_fn<1, false, LY::OFF_GAMMA, LY::OFF_BETA, LY::OFF_DUMMY>(lds, s, m0, wave, lane);
        CW_STAMP();   // 6 + 6 g: row pass
      } else {
        const __amdgpu_buffer_rsrc_t rsO = make_rsrc(s.out);
        const uint32_t vo = (uint32_t)(m0 + row) * (uint32_t)(s.ldo * 2);
        emit(sBias, s.oscale, [&](int cc, const u32x4_t& v) __attribute__((always_inline)) { __builtin_amdgcn_raw_buffer_store_b128(v, rsO, vo + cc * 2, 0, 0); });
        pend = 5;
        CW_STAMP();
        CW_STAMP();
      }
    }
  };
  if (lora_tile) tile_body(std::integral_constant<int, NSTG>{});
  else tile_body(std::integral_constant<int, 5>{});
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // trailing zero-fill DMAs still target the LDS
  CW_STAMP();   // drained
#undef CW_STAMP
}

}  // namespace aqlchain

namespace {

using namespace aqlchain;

// shared tail of the two entry points: argument checks of the filled descriptor, tile height, launch
int chain_launch(Args& a, long M, const char* name, bool bwd, hipStream_t stream, bool wide = false) {
  int nln = a.has_pre ? 1 : 0;
  bool nout_odd = false;   // a LayerNorm output that starts on an odd 64-row tile: 64-row tiles (the row pass decides per tile)
  for (int g = 0; g < a.nstage; ++g) {
    const Stage& s = a.st[g];
    AQL_CHECK_ARG(s.W != nullptr && s.ldw >= CH && (long)CH * s.ldw * 2 < (long)BUF_BYTES, "%s: stage %d has no weight", name, g);
    AQL_CHECK_ARG(s.Ad == nullptr || (s.Bup && s.T && s.Ts && a.S), "%s: stage %d: LoRA needs Bup, T, Ts and S", name, g);
    AQL_CHECK_ARG(s.keep || (s.out != nullptr && s.res == nullptr && !s.ln), "%s: stage %d: a DIRECT stage writes `out` only", name, g);
    AQL_CHECK_ARG(s.ln != RP_LN_FWD || (s.keep && s.gamma && s.beta && s.stats), "%s: stage %d: LayerNorm needs keep, gamma, beta, stats", name, g);
    AQL_CHECK_ARG(s.ln != RP_LN_BWD || (s.keep && s.gamma && s.stats && s.lnx && s.out), "%s: stage %d: LayerNorm backward needs keep, gamma, stats, x, out", name, g);
    AQL_CHECK_ARG(s.out == nullptr || (s.ldo >= CH && M * s.ldo * 2 < (long)BUF_BYTES), "%s: stage %d: output span", name, g);
    AQL_CHECK_ARG(s.res == nullptr || (s.ldr >= CH && M * s.ldr * 2 < (long)BUF_BYTES), "%s: stage %d: residual span", name, g);
    AQL_CHECK_ARG(s.lnx == nullptr || (s.ldlx >= CH && M * s.ldlx * 2 < (long)BUF_BYTES), "%s: stage %d: saved LayerNorm input span", name, g);
    AQL_CHECK_ARG(s.nout == nullptr || (s.ldn >= CH && M * s.ldn * 2 < (long)BUF_BYTES && s.nout_row0 % 64 == 0), "%s: stage %d: LayerNorm output span / first row (multiple of 64)", name, g);
    nout_odd = nout_odd || (s.nout != nullptr && s.nout_row0 % 128 != 0);
    nln += s.ln ? 1 : 0;
  }
  AQL_CHECK_ARG(nln <= 1, "%s: at most one LayerNorm per chain", name);
  static bool once = false;
  if (!once) {
    (void)hipFuncSetAttribute((const void*)chain_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, Lay<1>::TOTAL);
    (void)hipFuncSetAttribute((const void*)chain_kernel<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, Lay<2>::TOTAL);
    (void)hipFuncSetAttribute((const void*)chain_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, Lay<1>::TOTAL);
    (void)hipFuncSetAttribute((const void*)chain_wide_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LayW::TOTAL);
    once = true;
  }
#ifdef AQL_CHAIN_TRACE
  {
    const char* tb = getenv("AQL_CHAIN_TRACE_BUF");   // device address of the stamp buffer (tools/trace_chain.py, trace builds only)
    a.trace = tb ? (long long*)strtoull(tb, nullptr, 0) : nullptr;
  }
#else
  a.trace = nullptr;
#endif
  // 128-row tiles while they fill the chip (the twin forward: 256 tiles), 64-row tiles below that (the backward pass runs on the
  // watermarked half only: 16384 rows = 128 tiles of 128 -- half the CUs idle -- or 256 of 64)
  static const int force = AQL_TUNE_INT("AQL_CHAIN_BM", 0);   // tuning hook
  const bool small = force == 64 || (force == 0 && (M / 128 < 200 || M % 128 != 0 || a.rps % 128 != 0 || a.row0 % 128 != 0 || nout_odd));
  if (wide) hipLaunchKernelGGL(chain_wide_kernel, dim3((unsigned)(M / 64)), dim3(NTH), LayW::TOTAL, stream, a);   // rank 320: 64-row tiles
  else if (bwd) hipLaunchKernelGGL((chain_kernel<1, true>), dim3((unsigned)(M / 64)), dim3(NTH), Lay<1>::TOTAL, stream, a);   // (64-row tiles only)
  else if (small) hipLaunchKernelGGL((chain_kernel<1, false>), dim3((unsigned)(M / 64)), dim3(NTH), Lay<1>::TOTAL, stream, a);
  else hipLaunchKernelGGL((chain_kernel<2, false>), dim3((unsigned)(M / 128)), dim3(NTH), Lay<2>::TOTAL, stream, a);
  AQL_CHECK_LAUNCH(name);
  return AQL_OK;
}

}  // namespace

static int chain_fwd_common(bool wide, const bf16_t* X, long ldx, long M, int rows_per_sample, long lora_row0, const bf16_t* S, int nstage,
                            const void* const* W, const long* ldw, const void* const* bias, const void* const* Adown,
                            const void* const* Bup, void* const* T, void* const* Ts, const void* const* res, const long* ldr,
                            void* const* out, const long* ldo, const int* keep, const int* ln, const void* const* gamma,
                            const void* const* beta, const float* eps, void* const* stats, void* const* nout, const long* ldn,
                            const long* nout_row0, const float* oscale, hipStream_t stream) {
  AQL_CHECK_ARG(X != nullptr && nstage >= 1 && nstage <= MAXS, "aql_lora_chain_fwd: 1..%d stages", MAXS);
  AQL_CHECK_ARG(M > 0 && M % 64 == 0 && M < (1L << 30), "aql_lora_chain_fwd: M = %ld must be a multiple of 64", M);
  AQL_CHECK_ARG(rows_per_sample > 0 && rows_per_sample % 64 == 0, "aql_lora_chain_fwd: rows_per_sample %% 64 != 0");
  AQL_CHECK_ARG(lora_row0 >= 0 && lora_row0 % 64 == 0, "aql_lora_chain_fwd: lora_row0 %% 64 != 0");
  AQL_CHECK_ARG(ldx >= CH && (ldx % 8) == 0 && (long)M * ldx * 2 < (long)BUF_BYTES, "aql_lora_chain_fwd: input leading dimension / span");
  Args a;
  memset(&a, 0, sizeof(a));
  a.X = X;
  a.ldx = ldx;
  a.S = S;
  a.M = (int)M;
  a.rps = rows_per_sample;
  a.row0 = (int)lora_row0;
  a.nstage = nstage;
  for (int g = 0; g < nstage; ++g) {
    Stage& s = a.st[g];
    s.W = (const bf16_t*)W[g];
    s.ldw = ldw[g];
    s.bias = bias ? (const bf16_t*)bias[g] : nullptr;
    s.Ad = Adown ? (const bf16_t*)Adown[g] : nullptr;
    s.Bup = Bup ? (const bf16_t*)Bup[g] : nullptr;
    s.T = T ? (bf16_t*)T[g] : nullptr;
    s.Ts = Ts ? (bf16_t*)Ts[g] : nullptr;
    s.res = res ? (const bf16_t*)res[g] : nullptr;
    s.ldr = ldr ? ldr[g] : 0;
    s.out = out ? (bf16_t*)out[g] : nullptr;
    s.ldo = ldo ? ldo[g] : 0;
    s.keep = keep[g];
    s.ln = (ln && ln[g]) ? RP_LN_FWD : RP_NONE;
    s.gamma = gamma ? (const bf16_t*)gamma[g] : nullptr;
    s.beta = beta ? (const bf16_t*)beta[g] : nullptr;
    s.eps = eps ? eps[g] : 0.f;
    s.oscale = oscale ? oscale[g] : 1.f;
    s.stats = stats ? (float*)stats[g] : nullptr;
    s.nout = nout ? (bf16_t*)nout[g] : nullptr;
    s.ldn = ldn ? ldn[g] : 0;
    s.nout_row0 = nout_row0 ? (int)nout_row0[g] : 0;
  }
  return chain_launch(a, M, wide ? "aql_lora_chain_fwd_r320" : "aql_lora_chain_fwd", false, stream, wide);
}

extern "C" int aql_lora_chain_fwd(const bf16_t* X, long ldx, long M, int rows_per_sample, long lora_row0, const bf16_t* S, int nstage,
                                  const void* const* W, const long* ldw, const void* const* bias, const void* const* Adown,
                                  const void* const* Bup, void* const* T, void* const* Ts, const void* const* res, const long* ldr,
                                  void* const* out, const long* ldo, const int* keep, const int* ln, const void* const* gamma,
                                  const void* const* beta, const float* eps, void* const* stats, void* const* nout, const long* ldn,
                                  const long* nout_row0, const float* oscale, hipStream_t stream) {
  return chain_fwd_common(false, X, ldx, M, rows_per_sample, lora_row0, S, nstage, W, ldw, bias, Adown, Bup, T, Ts, res, ldr, out, ldo, keep,
                          ln, gamma, beta, eps, stats, nout, ldn, nout_row0, oscale, stream);
}

// The rank-320 form (chain_wide_kernel): same arguments; Adown[g] is [320][320] (rank x K), Bup[g] [320][320] (N x rank), T[g] / Ts[g]
// [M][320], S [M / rows_per_sample][320].
extern "C" int aql_lora_chain_fwd_r320(const bf16_t* X, long ldx, long M, int rows_per_sample, long lora_row0, const bf16_t* S, int nstage,
                                       const void* const* W, const long* ldw, const void* const* bias, const void* const* Adown,
                                       const void* const* Bup, void* const* T, void* const* Ts, const void* const* res, const long* ldr,
                                       void* const* out, const long* ldo, const int* keep, const int* ln, const void* const* gamma,
                                       const void* const* beta, const float* eps, void* const* stats, void* const* nout, const long* ldn,
                                       const long* nout_row0, const float* oscale, hipStream_t stream) {
  return chain_fwd_common(true, X, ldx, M, rows_per_sample, lora_row0, S, nstage, W, ldw, bias, Adown, Bup, T, Ts, res, ldr, out, ldo, keep,
                          ln, gamma, beta, eps, stats, nout, ldn, nout_row0, oscale, stream);
}

extern "C" int aql_lora_chain_bwd(const bf16_t* dY, long lddy, long M, int rows_per_sample, const bf16_t* S, int nstage,
                                  const void* const* Wt, const long* ldw, const void* const* BupT, const void* const* AT,
                                  void* const* dTs, void* const* dT, void* const* dX, const long* lddx, const int* keep,
                                  const void* const* ln_x, const long* ld_lnx, const void* const* ln_stats, const void* const* ln_gamma,
                                  const void* const* ln_dres, const long* ld_dres, void* const* ln_out, const long* ld_lnout,
                                  hipStream_t stream) {
  AQL_CHECK_ARG(dY != nullptr && nstage >= 1 && nstage <= MAXS, "aql_lora_chain_bwd: 1..%d stages", MAXS);
  AQL_CHECK_ARG(M > 0 && M % 64 == 0 && M < (1L << 30), "aql_lora_chain_bwd: M = %ld must be a multiple of 64", M);
  AQL_CHECK_ARG(rows_per_sample > 0 && rows_per_sample % 64 == 0, "aql_lora_chain_bwd: rows_per_sample %% 64 != 0");
  AQL_CHECK_ARG(lddy >= CH && (lddy % 8) == 0 && (long)M * lddy * 2 < (long)BUF_BYTES, "aql_lora_chain_bwd: input leading dimension / span");
  Args a;
  memset(&a, 0, sizeof(a));
  a.X = dY;
  a.ldx = lddy;
  a.S = S;
  a.M = (int)M;
  a.rps = rows_per_sample;
  a.row0 = 0;
  a.nstage = nstage;
  auto fill_ln = [&](Stage& s, int k) {   // entry k of the ln_* arrays: 0 = the pass on the chain input, g + 1 = behind stage g
    s.ln = RP_LN_BWD;
    s.lnx = (const bf16_t*)ln_x[k];
    s.ldlx = ld_lnx[k];
    s.stats = (float*)const_cast<void*>(ln_stats[k]);
    s.gamma = (const bf16_t*)ln_gamma[k];
    s.res = ln_dres ? (const bf16_t*)ln_dres[k] : nullptr;
    s.ldr = ld_dres ? ld_dres[k] : 0;
    s.out = (bf16_t*)ln_out[k];
    s.ldo = ld_lnout[k];
    s.keep = 1;
  };
  if (ln_x && ln_x[0]) {
    a.has_pre = 1;
    fill_ln(a.pre, 0);
    AQL_CHECK_ARG(a.pre.stats && a.pre.gamma && a.pre.out && a.pre.ldo >= CH && a.pre.ldlx >= CH && M * a.pre.ldo * 2 < (long)BUF_BYTES &&
                      M * a.pre.ldlx * 2 < (long)BUF_BYTES && (a.pre.res == nullptr || (a.pre.ldr >= CH && M * a.pre.ldr * 2 < (long)BUF_BYTES)),
                  "aql_lora_chain_bwd: the LayerNorm backward on the chain input needs x, stats, gamma, out (spans < 1 GiB)");
  }
  for (int g = 0; g < nstage; ++g) {
    Stage& s = a.st[g];
    s.W = (const bf16_t*)Wt[g];
    s.ldw = ldw[g];
    s.oscale = 1.f;
    s.Ad = BupT ? (const bf16_t*)BupT[g] : nullptr;
    s.Bup = AT ? (const bf16_t*)AT[g] : nullptr;
    s.T = dTs ? (bf16_t*)dTs[g] : nullptr;
    s.Ts = dT ? (bf16_t*)dT[g] : nullptr;
    s.keep = keep[g];
    if (s.keep && ln_x && ln_x[g + 1]) {
      fill_ln(s, g + 1);
    } else {
      s.out = dX ? (bf16_t*)dX[g] : nullptr;
      s.ldo = lddx ? lddx[g] : 0;
    }
  }
  return chain_launch(a, M, "aql_lora_chain_bwd", true, stream);
}
