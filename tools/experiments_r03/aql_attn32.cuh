// Self-attention forward on v_mfma_f32_32x32x16_bf16 with a cross-tile software pipeline (included by aql_attn.hip; same
// AttnArgs / layouts: q, k, v, o = the un-permuted linear outputs [B, N, H*d], head h owns columns [h*d, (h+1)*d)).
// Replaces F.scaled_dot_product_attention as reached from scripts/lib/original_unet.py:688-704 for the self-attention shapes
// (Nq % 128 == 0, Nk % 64 == 0, d = 40 | 80); everything else stays on the 16x16x32 kernels of aql_attn.hip.
//
// Why another formulation.  The 16x16x32 forward is issue-bound: per 64x64 tile a wavefront issues 56 MFMAs (912 cycles of
// matrix pipe) and ~370 VALU instructions (exp2, fma, max3, cvt_pk, rescale: ~1340 cycles), and measures 2530 -- the sum, not the
// maximum.  A SIMD issues one instruction per ~4 cycles whichever wavefront it comes from; only the issue slots that fall into
// the shadow of a running MFMA are free, and a 32-cycle 32x32x16 MFMA leaves 5-7 such slots where a 16-cycle 16x16x32 leaves 2-3
// (tools/micro/valu_rate.hip; MI355X_MICROARCH.md "single-issue instructions hidden per 32x32x16 gap").  To have VALU work to
// put there at all, the loop is rotated: the S = K.Q^T product of tile t+1 is issued together with the exponentials of tile t,
// and the row maximum of tile t+1 is taken under the P.V product of tile t.
//
// One workgroup = 4 wavefronts x 32 query rows.  Products are computed transposed, as in aql_attn.hip:
//   S^T[key][q] = K_tile (A: 32 keys x 16 d)   x  Q^T (B: 16 d x 32 q)       KS k-steps of 16 over d (d = 40 -> 48, 80 -> 80)
//   O^T[d][q]  += V_tile^T (A: 32 d x 16 keys) x  P^T (B: 16 keys x 32 q)    DB blocks of 32 over d  (d = 40 -> 64, 80 -> 96)
// so a query row's statistics live in ONE lane pair (lane j and j + 32) and the probabilities feed the second MFMA from the
// accumulator registers: C/D layout  col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)  means lane (j, h) holds
// keys 8g + 4h + e (g = reg >> 2, e = reg & 3) of a 32-key block; the B operand of k-step ks = 2 * block + s2 takes registers
// 8 * s2 .. 8 * s2 + 7, i.e. k-slot e8 <-> key 16 * ks + 8 * (e8 >> 2) + 4h + (e8 & 3), and the V^T fragments are gathered in the
// same order with ds_read_b64_tr_b16 (two 4-key x 16-column blocks per lane).
#pragma once

#ifndef A32_ABL
#define A32_ABL 0   // ablation build (tools/build_alt.sh): 1 no exp2, 2 no P.V, 4 no K.Q^T, 8 no staging / barriers, 16 no rescale, 32 no max
#endif
namespace a32 {

constexpr int TILE = 64;      // keys per streamed tile
constexpr int PITCH = 256;    // LDS bytes per tile row (16 chunks of 16 B; d <= 128)

// 16-byte chunk c of tile row r sits at chunk c ^ swz(r).  swz is a bijection of (r & 15) whose high two bits are r & 3: the
// 16 lanes the hardware groups in a ds_read_b128 of the K fragments (rows {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} at one
// logical chunk) hit 16 different physical chunks, and the 4 key rows x 4 chunks that one 32-lane half of a ds_read_b64_tr_b16
// gathers for a V^T fragment cover all 16 chunk positions once.
__device__ __forceinline__ int swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
__device__ __forceinline__ int toff(int row, int chunk) { return row * PITCH + ((chunk ^ swz(row)) << 4); }

// Register-prefetching stager of one streamed operand (the Stager of aql_attn.hip on this file's swizzle): slots are numbered
// over the LIVE chunks of the 64-row tile (row = id / (d/8), chunk = id % (d/8)); rows past the end re-read the last valid row.
template <int CH>   // CH = chunks per row of the zero-padded LDS image
struct Stager {
  static constexpr int NIT = (TILE * CH + 255) / 256;
  long p[NIT];
  uint4 v[NIT];
  int off[NIT];
  const bf16_t* g;
  long ld;
  int nrows, cl;
  __device__ __forceinline__ void point(int row0, int tid) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int id = tid + it * 256;
      const int row = id / cl, c = id - row * cl;
      const int r = min(row0 + row, nrows - 1);
      p[it] = off[it] >= 0 ? (long)r * ld + c * 8 : 0;
    }
  }
  __device__ __forceinline__ void init(char* lds, const bf16_t* g_, long ld_, int nrows_, int d, int tid) {
    g = g_, ld = ld_, nrows = nrows_, cl = d >> 3;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int id = tid + it * 256;
      const int row = id / cl, c = id - row * cl;
      off[it] = (id < TILE * cl) ? toff(row, c) : -1;
      const int prow = id / CH, pc = id - prow * CH;   // padding chunks d/8 .. CH: zeroed once, never overwritten
      if (id < TILE * CH && pc >= cl) *reinterpret_cast<uint4*>(lds + toff(prow, pc)) = make_uint4(0u, 0u, 0u, 0u);
    }
    point(0, tid);
  }
  __device__ __forceinline__ void fetch() {
#pragma unroll
    for (int it = 0; it < NIT; ++it) v[it] = *reinterpret_cast<const uint4*>(g + p[it]);   // unconditional (see aql_attn.hip)
  }
  __device__ __forceinline__ void commit(char* lds) {
#pragma unroll
    for (int it = 0; it < NIT; ++it)
      if (off[it] >= 0) *reinterpret_cast<uint4*>(lds + off[it]) = v[it];
  }
  __device__ __forceinline__ void next(int row0, int tid) {
    if (row0 + TILE <= nrows) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) p[it] += (long)TILE * ld;
    } else {
      point(row0, tid);
    }
  }
};

typedef short v4s_t __attribute__((ext_vector_type(4)));

// K fragments of one 64-key tile: kf[kb][s] = rows 32 kb + j, k-step s;  koff[s] = this lane's byte offset of k-step s
template <int KS>
__device__ __forceinline__ void load_k(bf16x8_t (&kf)[2][KS], const char* sK, const int (&koff)[KS]) {
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) kf[kb][ks] = *reinterpret_cast<const bf16x8_t*>(sK + kb * (32 * PITCH) + koff[ks]);
}
// S^T = K_tile x Q^T: the two 32-key blocks are independent accumulator chains, issued alternately (a dependent 32x32x16 MFMA
// cannot start before its predecessor's 16 passes have finished)
template <int KS>
__device__ __forceinline__ void qk_product(f32x16_t (&s)[2], const bf16x8_t (&kf)[2][KS], const bf16x8_t (&qf)[KS]) {
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      if (ks == 0) {
        f32x16_t z;
#pragma unroll
        for (int e = 0; e < 16; ++e) z[e] = 0.f;
        s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kb][ks], qf[ks], z, 0, 0, 0);
      } else {
        s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kb][ks], qf[ks], s[kb], 0, 0, 0);
      }
    }
  }
}
// V^T fragments of one 64-key tile: vf[ks][db] (k-step ks of 16 keys, block db of 32 columns), two transpose reads each
template <int DB>
__device__ __forceinline__ void load_v(bf16x8_t (&vf)[4][DB], const char* sV, const int (&voff)[DB][2]) {
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
#pragma unroll
    for (int db = 0; db < DB; ++db) {
      const char* base = sV + ks * (16 * PITCH);
      const v4s_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s_t __attribute__((address_space(3)))*)(base + voff[db][0]));
      const v4s_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s_t __attribute__((address_space(3)))*)(base + voff[db][1]));
      vf[ks][db] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
}

__device__ __forceinline__ float pair_max(float v) {   // max over the lane pair (j, j + 32)
  const uint32_t u = __float_as_uint(v);
  auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float pair_sum(float v) {
  const uint32_t u = __float_as_uint(v);
  auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// KS: k-steps of 16 over d for K.Q^T; DB: 32-wide blocks of d for P.V; CH: zero-padded chunks per LDS row (>= 2 KS, >= 4 DB).
// The first padding column of V (column d) holds 1.0: accumulator row d of O^T is the softmax denominator, summed from exactly
// the bf16 probabilities that multiply V (aql_attn.hip's ONES form).  Needs d < 32 DB.
template <int KS, int DB, int CH>
__global__ __launch_bounds__(256, 2) void attn_fwd32_kernel(const AttnArgs a) {
  __shared__ __attribute__((aligned(16))) char sK[TILE * PITCH];
  __shared__ __attribute__((aligned(16))) char sV[TILE * PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, hh = lane >> 5;
  int bx, h, b;
  attn_block(bx, h, b);
  const int q0 = bx * 128 + wave * 32;
  const bf16_t* qp = a.q + (long)b * a.Nq * a.ldq + h * a.d;
  const bf16_t* kp = a.k + (long)b * a.Nk * a.ldk + h * a.d;
  const bf16_t* vp = a.v + (long)b * a.Nk * a.ldv + h * a.d;
  // Q^T fragments (B operand): lane (j, hh) holds query row q0 + j, columns 16 s + 8 hh .. + 7
  bf16x8_t qf[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int col = s * 16 + hh * 8;
    const bool ok = col < a.d;
    const uint4 x = *reinterpret_cast<const uint4*>(qp + (long)(q0 + j) * a.ldq + (ok ? col : 0));
    uint4 v = mask4(x, ok);
    qf[s] = *reinterpret_cast<bf16x8_t*>(&v);
  }
  // per-lane LDS byte offsets: K fragment of k-step s (row j of a 32-key block; + 32 rows for the second block) ...
  int koff[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) koff[s] = toff(j, 2 * s + hh);
  // ... and the V^T gathers: 16-lane group g16 covers columns 32 db + 16 g16 .. + 15, lane p supplies row (p >> 2), columns
  // 4 (p & 3) ..; keys 16 ks + 4 hh + (p >> 2) (lo) and + 8 (hi).  swz of those rows does not depend on ks (16 ks = 0 mod 16).
  const int p16 = lane & 15, g16 = (lane >> 4) & 1;
  int voff[DB][2];
#pragma unroll
  for (int db = 0; db < DB; ++db)
#pragma unroll
    for (int hi = 0; hi < 2; ++hi)
      voff[db][hi] = toff(4 * hh + (p16 >> 2) + 8 * hi, 4 * db + 2 * g16 + ((p16 & 3) >> 1)) + (p16 & 1) * 8;

  f32x16_t o[DB];
#pragma unroll
  for (int db = 0; db < DB; ++db)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[db][e] = 0.f;
  float m = -INFINITY;
  const float c = a.scale * LOG2E;

  Stager<CH> stK, stV;
  stK.init(sK, kp, a.ldk, a.Nk, a.d, tid);
  stV.init(sV, vp, a.ldv, a.Nk, a.d, tid);
  stK.fetch();
  stV.fetch();
  __syncthreads();   // the padding chunks were zeroed by other threads
  if (tid < TILE) *reinterpret_cast<bf16_t*>(sV + toff(tid, a.d >> 3) + (a.d & 7) * 2) = (bf16_t)0x3F80;   // V[:, d] = 1.0
  stK.commit(sK);    // K(0)
  __syncthreads();
  stK.next(TILE, tid);
  stK.fetch();       // K(1) (a copy of the last tile's rows when there is none: never used)
  f32x16_t s[2];
  {
    bf16x8_t kf[2][KS];
    load_k<KS>(kf, sK, koff);
    qk_product<KS>(s, kf, qf);       // S(0)
  }

  const int nt = a.Nk / TILE;
  for (int t = 0; t < nt; ++t) {
    // here: s = raw scores of tile t; stV holds V(t), stK holds K(t+1)
#if !(A32_ABL & 8)
    __syncthreads();   // every wavefront is done with K(t) and V(t-1) in LDS
    stK.commit(sK);
    stV.commit(sV);
    __syncthreads();
    stK.next((t + 2) * TILE, tid);
    stK.fetch();
    stV.next((t + 1) * TILE, tid);
    stV.fetch();
#endif
    // all LDS fragment reads of the tile up front: an MFMA that waits for the ds_read issued just before it stalls the chain
    bf16x8_t kf[2][KS], vf[4][DB];
    load_k<KS>(kf, sK, koff);
    load_v<DB>(vf, sV, voff);
    // pin the global loads HERE: hipcc otherwise sinks them below the MFMAs (to the end of the iteration), and the commit at the
    // top of the next iteration then waits out the whole memory latency (+650 cycles per tile, measured)
    __builtin_amdgcn_sched_barrier(0);
    // row maximum of tile t (a query row = one lane pair)
#if A32_ABL & 32
    float mx = 0.f;
#else
    float mx = fmaxf(s[0][0], s[1][0]);
#pragma unroll
    for (int e = 1; e < 16; ++e) mx = fmaxf(fmaxf(mx, s[0][e]), s[1][e]);   // v_max3_f32 chains
    mx = pair_max(mx);
#endif
    const float mn = fmaxf(m, mx);
    const float alpha = __builtin_amdgcn_exp2f((m - mn) * c);
    m = mn;
    const float mnc = mn * c;
    // S(t+1) on the matrix pipe, the exponentials / packing / rescale of tile t in its shadow
    f32x16_t sn[2];
#if A32_ABL & 4
    sn[0] = s[0], sn[1] = s[1];
#else
    qk_product<KS>(sn, kf, qf);
#endif
    bf16x8_t pb[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int kb = ks >> 1, r0 = (ks & 1) * 8;
      float pv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
#if A32_ABL & 1
        pv[e] = __builtin_fmaf(s[kb][r0 + e], c, -mnc);
#else
        pv[e] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kb][r0 + e], c, -mnc));
#endif
      }
      uint4 w;
      w.x = pack_bf16x2(pv[0], pv[1]);
      w.y = pack_bf16x2(pv[2], pv[3]);
      w.z = pack_bf16x2(pv[4], pv[5]);
      w.w = pack_bf16x2(pv[6], pv[7]);
      pb[ks] = *reinterpret_cast<bf16x8_t*>(&w);
    }
#if !(A32_ABL & 16)
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int e = 0; e < 16; ++e) o[db][e] *= alpha;
#endif
    // O^T += V(t)^T . P^T
#pragma unroll
    for (int ks = 0; ks < ((A32_ABL & 2) ? 1 : 4); ++ks) {
#pragma unroll
      for (int db = 0; db < DB; ++db) o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[ks][db], pb[ks], o[db], 0, 0, 0);
    }
    s[0] = sn[0];
    s[1] = sn[1];
  }

  // denominator: accumulator row d of O^T -> block d >> 5, row i = d & 31 = (reg & 3) + 8 (reg >> 2) + 4 hh
  const int di = a.d & 31, ddb = a.d >> 5, dreg = ((di >> 3) << 2) | (di & 3), dh = (di >> 2) & 1;
  float lsum = 0.f;
#pragma unroll
  for (int db = 0; db < DB; ++db)
#pragma unroll
    for (int e = 0; e < 16; ++e)
      if (db == ddb && e == dreg && hh == dh) lsum = o[db][e];
  lsum = pair_sum(lsum);
  const float inv = 1.f / lsum;
  bf16_t* op = a.out + (long)b * a.Nq * a.ldo + h * a.d + (long)(q0 + j) * a.ldo;
#pragma unroll
  for (int db = 0; db < DB; ++db)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int col = 32 * db + 8 * g + 4 * hh;
      if (col < a.d)
        *reinterpret_cast<uint2*>(op + col) = make_uint2(pack_bf16x2(o[db][4 * g] * inv, o[db][4 * g + 1] * inv),
                                                         pack_bf16x2(o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv));
    }
  if (hh == 0) a.lse[((long)b * a.H + h) * a.Nq + q0 + j] = m * a.scale + logf(lsum);
}

}  // namespace a32
