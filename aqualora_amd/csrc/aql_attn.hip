// Flash-style scaled-dot-product attention (forward + backward) for the SD-1.5 U-Net head sizes d = 40/80/160,
// written for gfx950 wave64 + v_mfma_f32_16x16x32_bf16.  Replaces F.scaled_dot_product_attention as reached from
// diffusers' AttnProcessor2_0 / the in-tree twin scripts/lib/original_unet.py:688-704 (no mask, no dropout).
//
// Layout: q/k/v/o are the un-permuted linear outputs [B, N, H*d]; head h owns columns [h*d, (h+1)*d).
//
// All three kernels share one shape.  The workgroup's 4 wavefronts each OWN 32 rows (kept in registers as MFMA
// B-operands) and STREAM 64-row tiles of the other side through LDS:
//   "S-product"  acc[streamed 16-row frag][owner frag] += tile_rowmajor(A) x owner(B)    (contraction over d)
//   "T-product"  acc[d frag][owner frag]               += tile_transposed(A) x P(B)      (contraction over streamed rows)
// Products are computed transposed (streamed rows x owner rows) so that every softmax statistic of an owner row is
// lane-local up to a 4-lane-group exchange, and the probabilities feed the second MFMA straight from the
// accumulator registers: the MFMA k-slot order is permuted identically on both operands instead of shuffling.
//   forward : owner = Q,        stream K and V
//   dQ      : owner = Q, dO     stream K and V
//   dK/dV   : owner = K, V      stream Q, dO, LSE and delta
// Streamed tiles are staged ONCE, row-major; the T-product gathers its transposed fragments with the LDS
// transpose-read ds_read_b64_tr_b16, so there is no transposing staging pass and no second LDS image.
#include "aql_common.h"

namespace {

constexpr int TILE = 64;                    // streamed rows per tile
constexpr int OWN = 32;                     // owner rows per wavefront
constexpr float LOG2E = 1.4426950408889634f;

// LDS image of a streamed tile: [64 rows][pitch], pitch a multiple of the 256-B bank window, 16-B chunk index XORed with
// (row & 7) << 1.  With the hardware's real lane groups this is conflict-free for BOTH consumers: the ds_read_b128
// A-fragments of the S-product (group = rows {0-3,12-15} at chunk c + rows 4-11 at chunk c+1) and the 8-row x 32-B
// ds_read_b64_tr_b16 gathers of the T-product (bit 0 of the chunk is left alone so a 32-B pair stays a pair).
template <int DH>
struct RowPitch {
  static constexpr int value = (DH <= 128) ? 256 : 512;
};
template <int DH>
__device__ __forceinline__ int tile_off(int row, int chunk) {
  return row * RowPitch<DH>::value + ((chunk ^ ((row & 7) << 1)) << 4);
}

__device__ __forceinline__ uint4 zero4() { return make_uint4(0u, 0u, 0u, 0u); }
// AND-mask instead of a select: keeps the load unconditional (a select lets LLVM sink the load under a branch, and
// hipcc then waits vmcnt(0) after every such load)
__device__ __forceinline__ uint4 mask4(const uint4& v, bool ok) {
  const uint32_t m = 0u - (uint32_t)ok;
  return make_uint4(v.x & m, v.y & m, v.z & m, v.w & m);
}

// global [rows][ld] (head slice already applied to ptr) -> LDS row-major [64][DH] (zero padded)
template <int DH>
__device__ __forceinline__ void stage_rows(char* lds, const bf16_t* g, long ld, int row0, int nrows, int d, int tid) {
  constexpr int CPR = DH / 8;
  constexpr int NIT = (TILE * CPR + 255) / 256;
  uint4 v[NIT];
  // all loads first (unconditional, clamped address, masked after): branches would serialise them on vmcnt(0)
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int id = tid + it * 256;
    const int row = id / CPR, c = id - row * CPR;
    const bool ok = (id < TILE * CPR) & (row0 + row < nrows) & (c * 8 < d);
    const uint4 x = *reinterpret_cast<const uint4*>(g + (ok ? (long)(row0 + row) * ld + c * 8 : 0));
    v[it] = mask4(x, ok);
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int id = tid + it * 256;
    const int row = id / CPR, c = id - row * CPR;
    if (id < TILE * CPR) *reinterpret_cast<uint4*>(lds + tile_off<DH>(row, c)) = v[it];
  }
}

// Register-prefetching form of stage_rows for the streaming loops.  Per thread: NIT 16-byte slots of the 64-row tile.
//   init    : pointers for the first tile (rows clamped to the last valid row), LDS offsets, and the d..DH padding
//             chunks of the LDS image zeroed ONCE (they are never overwritten afterwards)
//   fetch   : issue the global loads of the current tile into registers (no masks, no branches)
//   commit  : registers -> LDS
//   next    : move to the tile starting at `row0`: a pointer bump for full tiles, a re-clamp for the last partial one
// Rows past the end of a partial tile are copies of the last valid row: finite values whose scores the callers mask
// (forward / dQ) or whose probabilities are zero through lse = +inf (dK/dV), so they never reach a result.
// The loop issues tile t+1's loads right after tile t is published in LDS, so their latency hides under the MFMAs and
// softmax of tile t instead of sitting between two barriers (the old stage_rows also spent ~70 VALU ops per tile on
// bounds masks and 64-bit address arithmetic in a loop whose VALU work already exceeds its MFMA work).
// PF = false (head sizes above 64, which already run at one workgroup per CU with AGPR-resident accumulators: the extra
// live registers cost more than the latency they hide, measured): same addressing, but the loads are issued inside
// commit(), i.e. synchronously between the two barriers.
template <int DH, bool PF = (DH <= 64)>
struct Stager {
  static constexpr int CPR = DH / 8;
  static constexpr int NIT = (TILE * CPR + 255) / 256;
  long p[NIT];   // element offset of the slot's 16 bytes from g (offsets, not pointers: pointer arrays end up in scratch)
  uint4 v[NIT];
  int off[NIT];  // LDS byte offset of the slot, -1: no slot (the tile has only TILE * d/8 live chunks)
  const bf16_t* g;
  long ld;
  int nrows, cl;  // cl = d / 8 live 16-byte chunks per row

  // Slots are numbered over the LIVE chunks only (row = id / cl, chunk = id % cl): at d = 40 a tile is 320 chunks, so
  // the second slot of wavefronts 1-3 is empty and skipped wave-uniformly instead of issuing a load + LDS write for
  // padding (staging measured at 57 of the 213 us of the d=40 forward with one slot per padded chunk).
  __device__ __forceinline__ void point(int row0, int tid) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int id = tid + it * 256;
      const int row = id / cl, c = id - row * cl;
      const int r = min(row0 + row, nrows - 1);
      p[it] = off[it] >= 0 ? (long)r * ld + c * 8 : 0;
    }
  }
  __device__ __forceinline__ void init(char* lds, const bf16_t* g_, long ld_, int row0, int nrows_, int d, int tid) {
    g = g_;
    ld = ld_;
    nrows = nrows_;
    cl = PF ? (d >> 3) : CPR;   // larger heads keep the compile-time mapping (their kernels are short: init cost matters)
    const int dl = d >> 3;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int id = tid + it * 256;
      const int row = id / cl, c = id - row * cl;
      off[it] = (id < TILE * cl && c < dl) ? tile_off<DH>(row, c) : -1;
      // padding chunks d/8 .. DH/8 of the LDS image: zeroed once, never overwritten
      const int prow = id / CPR, pc = id - prow * CPR;
      if (id < TILE * CPR && pc >= dl) *reinterpret_cast<uint4*>(lds + tile_off<DH>(prow, pc)) = zero4();
    }
    point(row0, tid);
  }
  __device__ __forceinline__ void load() {
#pragma unroll
    for (int it = 0; it < NIT; ++it) v[it] = *reinterpret_cast<const uint4*>(g + p[it]);  // unconditional: a branch
    // around a load makes hipcc wait vmcnt(0) right behind it; empty slots re-read g[0] (one cache line, all lanes)
  }
  __device__ __forceinline__ void fetch() {
    if constexpr (PF) load();
  }
  __device__ __forceinline__ void commit(char* lds) {
    if constexpr (!PF) load();
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

// LDS-DMA form of the streamed-tile staging, for the head sizes that run ONE workgroup per CU (DH > 64: d = 80 / 160, the 32x32 and
// 16x16 levels).  There the register-prefetch of Stager is off (its live registers cost more than they hide) and nothing else on the CU
// covers a tile's global-load latency: the 32x32 self-attention spent ~45 % of every 1.4 us tile waiting between its two barriers.
// buffer_load ... lds needs no staging registers: tile t+1 is requested into the OTHER LDS image at the top of iteration t and has
// landed when the iteration's single barrier is reached.  A wave instruction fills 1 KB of LDS = RPI whole image rows; lane (sub-row,
// physical chunk) fetches the logical chunk that the XOR swizzle of tile_off() maps there.  Lanes of padding chunks (>= d/8) never
// issue (EXEC-masked), so the zero padding and the ONES column written once at start-up survive; rows past the end of a partial tile
// are requested out of range (hardware zero fill) -- the callers mask them.
typedef unsigned int attn_u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t attn_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x40000000u, 0x00020000);
}
#ifndef AQL_ATTN_DMA_ABOVE
#define AQL_ATTN_DMA_ABOVE 64   // head sizes above this stage by LDS-DMA (experiment builds: 0 = all)
#endif
template <int DH>
struct DmaTile {
  static constexpr int PITCH = RowPitch<DH>::value;
  static constexpr int CPP = PITCH / 16;             // 16-byte chunks per image row (live + padding + unused)
  static constexpr int RPI = 1024 / PITCH;           // image rows per wave instruction
  static constexpr int NI = TILE / RPI / 4;          // instructions per wavefront and tile
  static constexpr int IMG = TILE * PITCH;
  __amdgpu_buffer_rsrc_t rs;
  uint32_t voff[NI];   // byte offset of the lane's 16 bytes inside tile 0 (row * ld + chunk), without the tile's first row
  int lrow[NI];        // the lane's row inside the tile, per instruction
  bool live;           // this lane carries a live chunk (same for every instruction: the chunk column only depends on the lane)
  uint32_t rowbytes;
  int nrows, blk0;
  __device__ __forceinline__ void init(const bf16_t* g, long ld, int nrows_, int d, int wave, int lane) {
    rs = attn_rsrc(g);
    rowbytes = (uint32_t)(ld * 2);
    nrows = nrows_;
    blk0 = wave;
    const int sub = lane / CPP, pc = lane % CPP;
    live = false;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int row = (i * 4 + wave) * RPI + sub;
      const int c = pc ^ ((row & 7) << 1);   // (row & 7) is the same for every i: rows of one lane differ by multiples of 4 * RPI = 8 or 16
      lrow[i] = row;
      voff[i] = (uint32_t)row * rowbytes + (uint32_t)c * 16u;
      live = (c * 8 < d);
    }
  }
  // request the tile whose first row is row0 into image `img` (asynchronous: covered by the issuing wavefront's vmcnt)
  __device__ __forceinline__ void issue(char* img, int row0) {
    const uint32_t soff = (uint32_t)row0 * rowbytes;
    if (live) {
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const uint32_t v = (row0 + lrow[i] < nrows) ? voff[i] : 0x80000000u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(img + ((i * 4 + blk0) << 10)), 16, v, soff, 0, 0);
      }
    }
  }
  // zero the padding chunks d/8 .. DH/8 of both images once (they are never written afterwards)
  __device__ __forceinline__ static void pad(char* imgs, int nimg, int d, int tid) {
    constexpr int CPR = DH / 8;
    const int dl = d >> 3;
    for (int id = tid; id < nimg * TILE * CPR; id += 256) {
      const int im = id / (TILE * CPR), r = id - im * TILE * CPR;
      const int row = r / CPR, pc = r - row * CPR;
      if (pc >= dl) *reinterpret_cast<uint4*>(imgs + im * IMG + tile_off<DH>(row, pc)) = zero4();
    }
  }
};

// owner rows -> B-operand fragments  f[frag][kstep]
template <int DH, int NOF>
__device__ __forceinline__ void load_owner(bf16x8_t (&f)[NOF][DH / 32], const bf16_t* g, long ld, int row0, int nrows,
                                           int d, int lane) {
#pragma unroll
  for (int fr = 0; fr < NOF; ++fr) {
    const int row = row0 + fr * 16 + (lane & 15);
#pragma unroll
    for (int s = 0; s < DH / 32; ++s) {
      const int col = s * 32 + (lane >> 4) * 8;
      const bool ok = (row < nrows) & (col < d);
      const uint4 x = *reinterpret_cast<const uint4*>(g + (ok ? (long)row * ld + col : 0));
      uint4 v = mask4(x, ok);
      f[fr][s] = *reinterpret_cast<bf16x8_t*>(&v);
    }
  }
}

// acc[sf][of] += rowmajor tile frag sf (A)  x  owner frag of (B)
// INIT: the accumulators start from zero -- the first k-step takes a literal-zero C operand instead of 16 x NOF v_mov per tile
template <int DH, int NOF, bool INIT = false>
__device__ __forceinline__ void s_product(f32x4_t (&acc)[4][NOF], const char* tile, const bf16x8_t (&own)[NOF][DH / 32],
                                          int lane) {
#pragma unroll
  for (int s = 0; s < DH / 32; ++s) {
    bf16x8_t a[4];
#pragma unroll
    for (int sf = 0; sf < 4; ++sf)
      a[sf] = *reinterpret_cast<const bf16x8_t*>(tile + tile_off<DH>(sf * 16 + (lane & 15), s * 4 + (lane >> 4)));
#pragma unroll
    for (int sf = 0; sf < 4; ++sf)
#pragma unroll
      for (int of = 0; of < NOF; ++of) {
        if (INIT && s == 0) acc[sf][of] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[sf], own[of][s], f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        else acc[sf][of] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[sf], own[of][s], acc[sf][of], 0, 0, 0);
      }
  }
}

// p[sf][of] (fp32, rows = streamed index sf*16 + g*4 + reg) -> B fragments over streamed rows:
// k-slot (g,e) of step s2  <->  streamed row 32*s2 + 16*(e>>2) + 4*g + (e&3)
template <int NOF>
__device__ __forceinline__ void pack_p(bf16x8_t (&pb)[2][NOF], const f32x4_t (&p)[4][NOF]) {
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
    for (int of = 0; of < NOF; ++of) {
      uint4 v;
      v.x = pack_bf16x2(p[2 * s2][of][0], p[2 * s2][of][1]);
      v.y = pack_bf16x2(p[2 * s2][of][2], p[2 * s2][of][3]);
      v.z = pack_bf16x2(p[2 * s2 + 1][of][0], p[2 * s2 + 1][of][1]);
      v.w = pack_bf16x2(p[2 * s2 + 1][of][2], p[2 * s2 + 1][of][3]);
      pb[s2][of] = *reinterpret_cast<bf16x8_t*>(&v);
    }
}

// acc[df][of] += (row-major tile)^T frag df (A)  x  pb (B).  The A fragment (i = column df*16 + lane&15 of the tile,
// k = 8 streamed rows) is gathered by the LDS transpose-read ds_read_b64_tr_b16: within a 16-lane group lane p supplies
// the address of row p>>2, columns 4*(p&3).. and receives column p of that 4x16 block (verified on hardware with
// tools/micro/tr_probe.hip).  k-slot (g,e) of step s2 <-> streamed row 32*s2 + 16*(e>>2) + 4*g + (e&3), as in pack_p.
typedef short v4s_t __attribute__((ext_vector_type(4)));
template <int DH, int DV, int NOF>
__device__ __forceinline__ void t_product(f32x4_t (&acc)[DV / 16][NOF], const char* tile, const bf16x8_t (&pb)[2][NOF],
                                          int lane) {
  const int p = lane & 15, g = lane >> 4;
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) {
    const int row = s2 * 32 + g * 4 + (p >> 2);  // rows row and row+16 share (row & 7): same swizzle
#pragma unroll
    for (int df = 0; df < DV / 16; ++df) {
      const char* base = tile + tile_off<DH>(row, df * 2 + ((p & 3) >> 1)) + (p & 1) * 8;
      const v4s_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s_t __attribute__((address_space(3)))*)(base));
      const v4s_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (v4s_t __attribute__((address_space(3)))*)(base + 16 * RowPitch<DH>::value));
      const bf16x8_t a = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
      for (int of = 0; of < NOF; ++of)
        acc[df][of] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, pb[s2][of], acc[df][of], 0, 0, 0);
    }
  }
}

// Reductions across the 4 lanes that share lane&15 (one per 16-lane row).  gfx950's v_permlane16_swap / v_permlane32_swap
// exchange whole rows / halves between two registers in one VALU op: with both operands equal to v the pair of results
// holds (v of the row-pair partner) in every lane, so xor-16 and xor-32 reductions cost two VALU ops each instead of a
// ds_bpermute round trip through the LDS pipe (~100+ cycles of dependent latency, four of them per K/V tile in the
// forward's online softmax).
__device__ __forceinline__ float group4_max(float v) {
  uint32_t u = __float_as_uint(v);
  auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  u = __float_as_uint(v);
  auto s = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__uint_as_float(s[0]), __uint_as_float(s[1]));
}
__device__ __forceinline__ float group4_sum(float v) {
  uint32_t u = __float_as_uint(v);
  auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  u = __float_as_uint(v);
  auto s = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(s[0]) + __uint_as_float(s[1]);
}

template <int N, int NOF>
__device__ __forceinline__ void zero_acc(f32x4_t (&a)[N][NOF]) {
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int j = 0; j < NOF; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) a[i][j][e] = 0.f;
}

// store acc^T (d x owner rows) as out[row][d] bf16, 4 consecutive d per lane
template <int DV, int NOF>
__device__ __forceinline__ void store_t(const f32x4_t (&acc)[DV / 16][NOF], bf16_t* g, long ld, int row0, int nrows, int d,
                                        const float (&muls)[NOF], int lane) {
#pragma unroll
  for (int of = 0; of < NOF; ++of) {
    const int row = row0 + of * 16 + (lane & 15);
    const float mul = muls[of];
    if (row >= nrows) continue;
#pragma unroll
    for (int df = 0; df < DV / 16; ++df) {
      const int col = df * 16 + (lane >> 4) * 4;
      if (col >= d) continue;
      *reinterpret_cast<uint2*>(g + (long)row * ld + col) =
          make_uint2(pack_bf16x2(acc[df][of][0] * mul, acc[df][of][1] * mul),
                     pack_bf16x2(acc[df][of][2] * mul, acc[df][of][3] * mul));
    }
  }
}

// same element mapping as store_t, fp32, no scaling (split-Q partials of dK/dV)
template <int DV>
__device__ __forceinline__ void store_t_f32(const f32x4_t (&acc)[DV / 16][2], float* g, long ld, int row0, int nrows, int d,
                                            int lane) {
#pragma unroll
  for (int of = 0; of < 2; ++of) {
    const int row = row0 + of * 16 + (lane & 15);
    if (row >= nrows) continue;
#pragma unroll
    for (int df = 0; df < DV / 16; ++df) {
      const int col = df * 16 + (lane >> 4) * 4;
      if (col >= d) continue;
      *reinterpret_cast<float4*>(g + (long)row * ld + col) =
          make_float4(acc[df][of][0], acc[df][of][1], acc[df][of][2], acc[df][of][3]);
    }
  }
}

struct AttnArgs {
  const bf16_t *q, *k, *v, *o, *dout;
  bf16_t *out, *dq, *dk, *dv;
  float* lse;
  float* delta;
  long ldq, ldk, ldv, ldo;
  long ldgq, ldgkv;   // backward: row stride of dq and of dk / dv (0 = dense, H * d; aql_sdpa_bwd_ex: column blocks of a wider buffer)
  int B, H, Nq, Nk, d;
  float scale;
  // exponent and natural-log factors of a raw score q.k:  p = exp2(s * cexp - ...), lse = max * cnat + log(sum).  (scale log2(e), scale) --
  // or (1, ln 2) when q arrives PRE-MULTIPLIED by scale log2(e) from its producer's epilogue (qpre: aql_sdpa_*_qpre; one bf16 rounding of
  // q c instead of q, so the no-FMA forward loop, FOLD = 2, keeps the precision of the default one).  dQ is the gradient of the UNSCALED
  // q in both cases (multiplier `scale`); dK = cnat dS^T q.
  float cexp, cnat;
  int qpre;
  int qsplit;    // dK/dV only: the streamed Q range is cut into qsplit pieces (grid.z = B * qsplit) ...
  float* part;   // ... whose fp32 partial results [2][qsplit][B][H][Nk][d] are summed by attn_dkv_reduce_kernel
};

// Occupancy hint (second __launch_bounds__ argument): with >= 2 workgroups per CU the register budget is 256 VGPRs and
// LLVM selects the VGPR form of the MFMAs.  Without it the accumulators live in AGPRs and every softmax / rescale step
// round-trips them through v_accvgpr_read/write: ~190 extra VALU moves per K/V tile in a loop whose VALU work
// (exp2, max, sum, pack) already outweighs its 28 MFMAs.  Only the head sizes that fit 256 registers without spilling
// get the hint.
// XCD-aware block order: hardware block L (x fastest) runs on XCD L % 8, each with a private L2.  All row blocks of one
// (sample, head) stream the SAME K/V (or Q/dO) panels, so they are given to one XCD: logical = (L % 8) * (total / 8) + L / 8
// (a bijection when the grid is a multiple of 8; otherwise the identity).  Returns (row block, head, sample).
__device__ __forceinline__ void attn_block(int& bx, int& h, int& b) {
  const int nx = gridDim.x, ny = gridDim.y, total = nx * ny * gridDim.z;
  int L = blockIdx.x + nx * (blockIdx.y + ny * blockIdx.z);
  if ((total & 7) == 0) L = (L & 7) * (total >> 3) + (L >> 3);
  bx = L % nx;
  const int r = L / nx;
  h = r % ny;
  b = r / ny;
}

// ------------------------------------------------------------------------------------------------ forward
// NOF = owner fragments (16 query rows each) per wavefront: 2 (32 rows, 128 per workgroup) or 4 (64 rows, 256 per
// workgroup).  With 4 the K/V staging, the barriers and the LDS fragment reads of a tile are amortised over twice the
// MFMA work and the two independent row halves give the scheduler something to overlap with the softmax chain.
// ONES (needs d < DV, i.e. a padding column in the V tile): the first padding column of V is set to 1.0 once, so the P.V
// product also yields the softmax denominator sum_j p_ij in accumulator column d -- the 16 packed adds, the two permlane
// reductions and the running-sum update per row block and tile disappear from the VALU stream, and the denominator is the
// sum of exactly the bf16 probabilities that multiply V.
// FOLD (round 4, needs ONES): the loop keeps NO running maximum.  The shift of a row is fixed by the FIRST tile (its row maximum) and
// never updated: later scores above it give probabilities above 1, which fp32 / bf16 carry as well as values below 1, and the
// accumulator rescale disappears with the update.  What a running maximum guards against is overflow -- a score ~2^127 above the
// first tile's maximum -- and that is detected at the end (non-finite or zero denominator / output): the workgroup then re-runs the
// classic pass (tools/probe_ops.py forces it; never seen on the U-Net's data).  Unlike the thresholded rescale tried in round 3,
// nothing here depends on a data-dependent decision after tile 0.
//   FOLD = 1 (default):  p = exp2(fma(s, c, -m c)) on pairs (v_pk_fma_f32; measured equal to the scalar FMA: fp32 FMA already issues 32 lanes per clock) -- the arithmetic of the classic loop, same precision;
//                        per streamed element  pk_fma/2 + exp2 + cvt/2  instead of  max3/2 + fma + exp2 + cvt/2 + rescale.
//   FOLD = 2 (opt-in, AQL_ATTN_FOLD=2; needs a spare K column, d < DH): the shift rides in the S-product -- Q is scaled by
//                        scale * log2(e) once (bf16), the first padding column of the K tile is 1.0 and the same column of a Q row
//                        carries -M_row, an INTEGER (exact in bf16; an integer shift scales probabilities, denominator and O by the
//                        same exact power of two, so the normalised output does not depend on which integer it is).  No FMA at all
//                        (278 -> 219 us at 8 x 8 x 4096^2 x 40), but the second bf16 rounding of q c costs ~0.0002 |s| log2 units
//                        in the scores: errors double on typical data and reach 1.7e-2 at |s| ~ 100 (profiles/r04_attention_fold.txt).
template <int DH, int DV, int NOF, bool ONES, int FOLD>
__device__ __forceinline__ bool attn_fwd_pass(const AttnArgs& a, char* sK, char* sV, int IMG_, int bx, int h, int b,
                                              f32x4_t (&o)[DV / 16][NOF], float (&inv)[NOF], float (&lse)[NOF]) {
  constexpr bool DMA = DH > AQL_ATTN_DMA_ABOVE;                      // one workgroup per CU: LDS-DMA into two images, one barrier per tile
  constexpr int IMG = TILE * RowPitch<DH>::value;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q0 = bx * (64 * NOF) + wave * (16 * NOF);
  const bf16_t* qp = a.q + (long)b * a.Nq * a.ldq + h * a.d;
  const bf16_t* kp = a.k + (long)b * a.Nk * a.ldk + h * a.d;
  const bf16_t* vp = a.v + (long)b * a.Nk * a.ldv + h * a.d;
  bf16x8_t qf[NOF][DH / 32];
  load_owner<DH>(qf, qp, a.ldq, q0, a.Nq, a.d, lane);
  const float c = a.cexp;
  uint4 cm[DH / 32];   // FOLD = 2: halfword mask of the shift column in this lane's Q fragments
  if constexpr (FOLD == 2) {
#pragma unroll
    for (int fr = 0; fr < NOF; ++fr)
#pragma unroll
      for (int ks = 0; ks < DH / 32; ++ks) {
        uint4 v = *reinterpret_cast<uint4*>(&qf[fr][ks]);
        v.x = pack_bf16x2(bf16lo(v.x) * c, bf16hi(v.x) * c);
        v.y = pack_bf16x2(bf16lo(v.y) * c, bf16hi(v.y) * c);
        v.z = pack_bf16x2(bf16lo(v.z) * c, bf16hi(v.z) * c);
        v.w = pack_bf16x2(bf16lo(v.w) * c, bf16hi(v.w) * c);
        qf[fr][ks] = *reinterpret_cast<bf16x8_t*>(&v);
      }
    const int e = a.d & 7;
    const uint32_t hw = (e & 1) ? 0xffff0000u : 0x0000ffffu;
#pragma unroll
    for (int ks = 0; ks < DH / 32; ++ks) {
      const bool mine = (ks == (a.d >> 5)) & ((lane >> 4) == ((a.d & 31) >> 3));
      const uint32_t mk = mine ? hw : 0u;
      cm[ks] = make_uint4((e >> 1) == 0 ? mk : 0u, (e >> 1) == 1 ? mk : 0u, (e >> 1) == 2 ? mk : 0u, (e >> 1) == 3 ? mk : 0u);
    }
  }
  zero_acc(o);
  float m[NOF], l[NOF];
#pragma unroll
  for (int of = 0; of < NOF; ++of) m[of] = FOLD ? 0.f : -INFINITY, l[of] = 0.f;
  float mc[NOF];       // FOLD = 1: the row's fixed shift m * c
#pragma unroll
  for (int of = 0; of < NOF; ++of) mc[of] = 0.f;
  Stager<DH> stK, stV;
  DmaTile<DH> dmK, dmV;
  if constexpr (DMA) {
    dmK.init(kp, a.ldk, a.Nk, a.d, wave, lane);
    dmV.init(vp, a.ldv, a.Nk, a.d, wave, lane);
    DmaTile<DH>::pad(sK, 2, a.d, tid);
    DmaTile<DH>::pad(sV, 2, a.d, tid);
    dmK.issue(sK, 0);
    dmV.issue(sV, 0);
  } else {
    stK.init(sK, kp, a.ldk, 0, a.Nk, a.d, tid);
    stV.init(sV, vp, a.ldv, 0, a.Nk, a.d, tid);
    stK.fetch();
    stV.fetch();
  }
  if constexpr (ONES || FOLD == 2) {
    __syncthreads();  // the padding chunks were zeroed by other threads
    if (tid < TILE) {
#pragma unroll
      for (int i = 0; i < (DMA ? 2 : 1); ++i) {
        if constexpr (ONES) *reinterpret_cast<bf16_t*>(sV + i * IMG + tile_off<DH>(tid, a.d >> 3) + (a.d & 7) * 2) = (bf16_t)0x3F80;  // 1.0
        if constexpr (FOLD == 2) *reinterpret_cast<bf16_t*>(sK + i * IMG + tile_off<DH>(tid, a.d >> 3) + (a.d & 7) * 2) = (bf16_t)0x3F80;
      }
    }
  }
  if constexpr (DMA) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // tile 0 has landed in image 0 (every wavefront waited for its own requests)
  }
  int cur = 0;
  for (int kt = 0; kt < a.Nk; kt += TILE) {
    const char* tK = sK;
    const char* tV = sV;
    if constexpr (DMA) {
      tK = sK + cur * IMG;
      tV = sV + cur * IMG;
      if (kt + TILE < a.Nk) {   // tile t+1 flies into the other image under this tile's MFMAs and softmax
        dmK.issue(sK + (cur ^ 1) * IMG, kt + TILE);
        dmV.issue(sV + (cur ^ 1) * IMG, kt + TILE);
      }
    } else {
      __syncthreads();
      stK.commit(sK);
      stV.commit(sV);
      __syncthreads();
      if (kt + TILE < a.Nk) {  // next tile's loads fly under this tile's MFMAs and softmax
        stK.next(kt + TILE, tid);
        stV.next(kt + TILE, tid);
        stK.fetch();
        stV.fetch();
      }
    }
    f32x4_t s[4][NOF];
    s_product<DH, NOF, true>(s, tK, qf, lane);
    if (kt + TILE > a.Nk) {
#pragma unroll
      for (int sf = 0; sf < 4; ++sf)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (kt + sf * 16 + (lane >> 4) * 4 + e >= a.Nk) {
#pragma unroll
            for (int of = 0; of < NOF; ++of) s[sf][of][e] = -INFINITY;
          }
    }
    if constexpr (FOLD == 1) {
      if (kt == 0) {   // the shift: the first tile's row maximum, fixed from here on
#pragma unroll
        for (int of = 0; of < NOF; ++of) {
          float mx = -INFINITY;
#pragma unroll
          for (int sf = 0; sf < 4; ++sf)
#pragma unroll
            for (int e = 0; e < 4; ++e) mx = fmaxf(mx, s[sf][of][e]);
          m[of] = group4_max(mx);
          mc[of] = m[of] * c;
        }
      }
#pragma unroll
      for (int of = 0; of < NOF; ++of) {
        const aql_f32x2_t c2 = aql_splat2(c), nm2 = aql_splat2(-mc[of]);
#pragma unroll
        for (int sf = 0; sf < 4; ++sf) {
          const aql_f32x2_t x0 = __builtin_elementwise_fma(aql_f32x2_t{s[sf][of][0], s[sf][of][1]}, c2, nm2);
          const aql_f32x2_t x1 = __builtin_elementwise_fma(aql_f32x2_t{s[sf][of][2], s[sf][of][3]}, c2, nm2);
          s[sf][of][0] = __builtin_amdgcn_exp2f(x0.x);
          s[sf][of][1] = __builtin_amdgcn_exp2f(x0.y);
          s[sf][of][2] = __builtin_amdgcn_exp2f(x1.x);
          s[sf][of][3] = __builtin_amdgcn_exp2f(x1.y);
        }
      }
    } else if constexpr (FOLD == 2) {
      if (kt == 0) {   // the shift: integer ceiling of the first tile's row maximum (bf16), applied here by hand and from now on by the MFMA
#pragma unroll
        for (int of = 0; of < NOF; ++of) {
          float mx = -INFINITY;
#pragma unroll
          for (int sf = 0; sf < 4; ++sf)
#pragma unroll
            for (int e = 0; e < 4; ++e) mx = fmaxf(mx, s[sf][of][e]);
          mx = group4_max(mx);
          const uint32_t mb = pack_bf16x2(ceilf(mx), 0.f) & 0xffffu;
          const float mf = bf16lo(mb);
          m[of] = mf;
          const uint32_t neg = (mb ^ 0x8000u) * 0x00010001u;   // -M in both halfwords
#pragma unroll
          for (int ks = 0; ks < DH / 32; ++ks) {
            uint4 v = *reinterpret_cast<uint4*>(&qf[of][ks]);
            v.x = (v.x & ~cm[ks].x) | (neg & cm[ks].x);
            v.y = (v.y & ~cm[ks].y) | (neg & cm[ks].y);
            v.z = (v.z & ~cm[ks].z) | (neg & cm[ks].z);
            v.w = (v.w & ~cm[ks].w) | (neg & cm[ks].w);
            qf[of][ks] = *reinterpret_cast<bf16x8_t*>(&v);
          }
#pragma unroll
          for (int sf = 0; sf < 4; ++sf)
#pragma unroll
            for (int e = 0; e < 4; ++e) s[sf][of][e] = __builtin_amdgcn_exp2f(s[sf][of][e] - mf);
        }
      } else {
#pragma unroll
        for (int of = 0; of < NOF; ++of)
#pragma unroll
          for (int sf = 0; sf < 4; ++sf)
#pragma unroll
            for (int e = 0; e < 4; ++e) s[sf][of][e] = __builtin_amdgcn_exp2f(s[sf][of][e]);
      }
    } else {
#pragma unroll
      for (int of = 0; of < NOF; ++of) {
        float mx = -INFINITY;
#pragma unroll
        for (int sf = 0; sf < 4; ++sf)
#pragma unroll
          for (int e = 0; e < 4; ++e) mx = fmaxf(mx, s[sf][of][e]);
        mx = group4_max(mx);
        // (Deferring the rescale until the maximum grows by 2^8 -- guide T13 -- measured -2...4 % on this kernel, but makes the
        // bf16 rounding of P depend on a threshold decision: a 1e-7 input perturbation then moves outputs by a bf16 ulp
        // everywhere instead of nowhere, which the sampler-vs-restatement test (1e-4 over 5 guided steps) rightly rejects.)
        const float mn = fmaxf(m[of], mx);
        const float alpha = __builtin_amdgcn_exp2f((m[of] - mn) * c);
        m[of] = mn;
        const float mnc = mn * c;
        float rs = 0.f;
#pragma unroll
        for (int sf = 0; sf < 4; ++sf)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[sf][of][e], c, -mnc));  // one FMA + one v_exp_f32
            s[sf][of][e] = p;
            if constexpr (!ONES) rs += p;
          }
        if constexpr (!ONES) l[of] = l[of] * alpha + group4_sum(rs);
#pragma unroll
        for (int df = 0; df < DV / 16; ++df)
#pragma unroll
          for (int e = 0; e < 4; ++e) o[df][of][e] *= alpha;
      }
    }
    bf16x8_t pb[2][NOF];
    pack_p(pb, s);
    t_product<DH, DV>(o, tV, pb, lane);
    if constexpr (DMA) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wavefront's share of tile t+1 has landed
      __syncthreads();   // everyone's has, and nobody reads image `cur` any more
      cur ^= 1;
    }
  }
  if constexpr (ONES) {  // the denominator of row (lane & 15) sits in accumulator column d: fragment d/16, lane group (d%16)/4
    const int df = a.d >> 4, grp = (a.d & 15) >> 2, e = a.d & 3;
#pragma unroll
    for (int of = 0; of < NOF; ++of) {
      float v = 0.f;
#pragma unroll
      for (int x = 0; x < DV / 16; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y)
          if (x == df && y == e && (lane >> 4) == grp) v = o[x][of][y];
      l[of] = group4_sum(v);
    }
  }
  bool ok = true;
#pragma unroll
  for (int of = 0; of < NOF; ++of) {
    inv[of] = 1.f / l[of];
    if constexpr (FOLD != 0) {
      lse[of] = (FOLD == 2 ? m[of] * 0.6931471805599453f : m[of] * a.cnat) + logf(l[of]);
      float big = 0.f;
#pragma unroll
      for (int df = 0; df < DV / 16; ++df)
#pragma unroll
        for (int e = 0; e < 4; ++e) big = fmaxf(big, fabsf(o[df][of][e]));
      ok &= (l[of] > 0.f) & (l[of] < INFINITY) & (big < INFINITY);    // NaN fails every comparison
    } else {
      lse[of] = m[of] * a.cnat + logf(l[of]);
    }
  }
  return ok;
}

template <int DH, int DV, int NOF, bool ONES, int FOLD = 0>
__global__ __launch_bounds__(256, (DH <= 64 ? 2 : 1)) void attn_fwd_kernel(const AttnArgs a) {
  constexpr bool DMA = DH > AQL_ATTN_DMA_ABOVE;
  constexpr int IMG = TILE * RowPitch<DH>::value;
  __shared__ __attribute__((aligned(1024))) char sK[(DMA ? 2 : 1) * IMG];
  __shared__ __attribute__((aligned(1024))) char sV[(DMA ? 2 : 1) * IMG];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int bx, h, b;
  attn_block(bx, h, b);
  const int q0 = bx * (64 * NOF) + wave * (16 * NOF);
  f32x4_t o[DV / 16][NOF];
  float inv[NOF], lse[NOF];
  bool ok = attn_fwd_pass<DH, DV, NOF, ONES, FOLD>(a, sK, sV, IMG, bx, h, b, o, inv, lse);
  if constexpr (FOLD != 0) {
    if (__syncthreads_or(!ok)) attn_fwd_pass<DH, DV, NOF, ONES, 0>(a, sK, sV, IMG, bx, h, b, o, inv, lse);   // overflow: the classic pass
  }
  store_t<DV>(o, a.out + (long)b * a.Nq * a.ldo + h * a.d, a.ldo, q0, a.Nq, a.d, inv, lane);
  if ((lane >> 4) == 0) {
#pragma unroll
    for (int of = 0; of < NOF; ++of) {
      const int row = q0 + of * 16 + (lane & 15);
      if (row < a.Nq) a.lse[((long)b * a.H + h) * a.Nq + row] = lse[of];
    }
  }
}

// ------------------------------------------------------------------------------------------------ dQ
#ifndef AQL_ATTN_DQ_OCC2_UPTO
#define AQL_ATTN_DQ_OCC2_UPTO 96   // head sizes up to this are compiled for two workgroups per CU (d = 80: 284 -> 250 registers, no spill; 80.1 -> 78.5 us backward at 4 x 1024 x 8 x 80)
#endif
// DFOLD (round 4; needs two spare head columns, d < DH): `dP - delta` comes out of the dP product.  delta is constant along a query row,
// so -delta rides as a (hi, lo) bf16 pair in the first two padding columns of the row's dO fragment against two columns of 1.0 in the
// streamed V tile: the fp32 accumulation adds it exactly, delta is represented to 2^-17, and one of the five VALU operations per
// element leaves the loop.
// LFOLD (round 5; q pre-multiplied by scale log2(e), aql_sdpa_bwd_qpre; same two spare columns, of Q and K this time): the row's
// -lse log2(e) rides as a (hi, lo) bf16 pair in the padding columns of its q fragment against two columns of 1.0 in the streamed K tile, so
// the S-product already is the exponent: p = exp2(s), the multiply-add per score leaves the loop (bound measured beforehand: -6.5 % of
// the two backward kernels at d = 40, profiles/r05_attention_shift_fma_bound.txt).  lse is represented to 2^-17.
template <int DH, int DV, bool DFOLD = false, bool LFOLD = false>
__global__ __launch_bounds__(256, (DH <= AQL_ATTN_DQ_OCC2_UPTO ? 2 : 1)) void attn_dq_kernel(const AttnArgs a) {
  constexpr bool DMA = DH > AQL_ATTN_DMA_ABOVE;   // see attn_fwd_kernel / DmaTile
  constexpr int IMG = TILE * RowPitch<DH>::value;
  __shared__ __attribute__((aligned(1024))) char sK[(DMA ? 2 : 1) * IMG];
  __shared__ __attribute__((aligned(1024))) char sV[(DMA ? 2 : 1) * IMG];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bx, h, b;
  attn_block(bx, h, b);
  const int q0 = bx * (4 * OWN) + wave * OWN;
  const bf16_t* qp = a.q + (long)b * a.Nq * a.ldq + h * a.d;
  const bf16_t* dop = a.dout + (long)b * a.Nq * a.ldo + h * a.d;
  const bf16_t* kp = a.k + (long)b * a.Nk * a.ldk + h * a.d;
  const bf16_t* vp = a.v + (long)b * a.Nk * a.ldv + h * a.d;
  bf16x8_t qf[2][DH / 32], dof[2][DH / 32];
  load_owner<DH>(qf, qp, a.ldq, q0, a.Nq, a.d, lane);
  load_owner<DH>(dof, dop, a.ldo, q0, a.Nq, a.d, lane);
  // delta[row] = sum_d dO[row,d] * O[row,d], computed here from the owner fragments (each lane holds 8 columns per
  // k-step of its row; the 4 lane groups of a row are summed) instead of a separate kernel; written out for dK/dV.
  float lse2[2], dl[2];
  {
    bf16x8_t ofr[2][DH / 32];
    load_owner<DH>(ofr, a.o + (long)b * a.Nq * a.ldo + h * a.d, a.ldo, q0, a.Nq, a.d, lane);
#pragma unroll
    for (int of = 0; of < 2; ++of) {
      float acc = 0.f;
#pragma unroll
      for (int s = 0; s < DH / 32; ++s) {
        const uint4 x = *reinterpret_cast<const uint4*>(&dof[of][s]);
        const uint4 y = *reinterpret_cast<const uint4*>(&ofr[of][s]);
        acc += bf16lo(x.x) * bf16lo(y.x) + bf16hi(x.x) * bf16hi(y.x) + bf16lo(x.y) * bf16lo(y.y) + bf16hi(x.y) * bf16hi(y.y) +
               bf16lo(x.z) * bf16lo(y.z) + bf16hi(x.z) * bf16hi(y.z) + bf16lo(x.w) * bf16lo(y.w) + bf16hi(x.w) * bf16hi(y.w);
      }
      dl[of] = group4_sum(acc);
    }
  }
#pragma unroll
  for (int of = 0; of < 2; ++of) {
    const int row = q0 + of * 16 + (lane & 15);
    const bool ok = row < a.Nq;
    lse2[of] = ok ? a.lse[((long)b * a.H + h) * a.Nq + row] * LOG2E : (LFOLD ? 1e30f : INFINITY);   // (rows past the end: p = 0)
    if (ok && (lane >> 4) == 0) a.delta[((long)b * a.H + h) * a.Nq + row] = dl[of];
  }
  if constexpr (LFOLD) {   // columns d, d + 1 of the row's q fragment: (-lse2_hi, -lse2_lo)
#pragma unroll
    for (int of = 0; of < 2; ++of) {
      const uint32_t hi = pack_bf16x2(lse2[of], 0.f) & 0xffffu;
      const uint32_t lo = pack_bf16x2(lse2[of] - bf16lo(hi), 0.f) & 0xffffu;
      const uint32_t word = (hi ^ 0x8000u) | ((lo ^ 0x8000u) << 16);
#pragma unroll
      for (int ks = 0; ks < DH / 32; ++ks)
        if (ks == (a.d >> 5) && (lane >> 4) == ((a.d & 31) >> 3)) {
          uint4 v = *reinterpret_cast<uint4*>(&qf[of][ks]);
          v.x = word;
          qf[of][ks] = *reinterpret_cast<bf16x8_t*>(&v);
        }
    }
  }
  if constexpr (DFOLD) {   // columns d, d + 1 of the row's dO fragment: (-delta_hi, -delta_lo); d % 8 == 0: word 0 of chunk d / 8
#pragma unroll
    for (int of = 0; of < 2; ++of) {
      const uint32_t hi = pack_bf16x2(dl[of], 0.f) & 0xffffu;
      const uint32_t lo = pack_bf16x2(dl[of] - bf16lo(hi), 0.f) & 0xffffu;
      const uint32_t word = (hi ^ 0x8000u) | ((lo ^ 0x8000u) << 16);
#pragma unroll
      for (int ks = 0; ks < DH / 32; ++ks)
        if (ks == (a.d >> 5) && (lane >> 4) == ((a.d & 31) >> 3)) {
          uint4 v = *reinterpret_cast<uint4*>(&dof[of][ks]);
          v.x = word;
          dof[of][ks] = *reinterpret_cast<bf16x8_t*>(&v);
        }
    }
  }
  f32x4_t dq[DV / 16][2];
  zero_acc(dq);
  const float c = a.cexp;
  Stager<DH> stK, stV;
  DmaTile<DH> dmK, dmV;
  if constexpr (DMA) {
    dmK.init(kp, a.ldk, a.Nk, a.d, wave, lane);
    dmV.init(vp, a.ldv, a.Nk, a.d, wave, lane);
    DmaTile<DH>::pad(sK, 2, a.d, tid);
    DmaTile<DH>::pad(sV, 2, a.d, tid);
    if constexpr (DFOLD || LFOLD) {
      __syncthreads();   // the padding chunks were zeroed by other threads
      if (tid < TILE) {
        if constexpr (DFOLD) {
          *reinterpret_cast<uint32_t*>(sV + tile_off<DH>(tid, a.d >> 3)) = 0x3F803F80u;         // 1.0 | 1.0
          *reinterpret_cast<uint32_t*>(sV + IMG + tile_off<DH>(tid, a.d >> 3)) = 0x3F803F80u;
        }
        if constexpr (LFOLD) {
          *reinterpret_cast<uint32_t*>(sK + tile_off<DH>(tid, a.d >> 3)) = 0x3F803F80u;
          *reinterpret_cast<uint32_t*>(sK + IMG + tile_off<DH>(tid, a.d >> 3)) = 0x3F803F80u;
        }
      }
    }
    dmK.issue(sK, 0);
    dmV.issue(sV, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  } else {
    stK.init(sK, kp, a.ldk, 0, a.Nk, a.d, tid);
    stV.init(sV, vp, a.ldv, 0, a.Nk, a.d, tid);
    if constexpr (DFOLD || LFOLD) {
      __syncthreads();   // the padding chunks were zeroed by other threads; they are never overwritten afterwards
      if (tid < TILE) {
        if constexpr (DFOLD) *reinterpret_cast<uint32_t*>(sV + tile_off<DH>(tid, a.d >> 3)) = 0x3F803F80u;
        if constexpr (LFOLD) *reinterpret_cast<uint32_t*>(sK + tile_off<DH>(tid, a.d >> 3)) = 0x3F803F80u;
      }
    }
    stK.fetch();
    stV.fetch();
  }
  int cur = 0;
  for (int kt = 0; kt < a.Nk; kt += TILE) {
    const char* tK = sK;
    const char* tV = sV;
    if constexpr (DMA) {
      tK = sK + cur * IMG;
      tV = sV + cur * IMG;
      if (kt + TILE < a.Nk) {
        dmK.issue(sK + (cur ^ 1) * IMG, kt + TILE);
        dmV.issue(sV + (cur ^ 1) * IMG, kt + TILE);
      }
    } else {
      __syncthreads();
      stK.commit(sK);
      stV.commit(sV);
      __syncthreads();
      if (kt + TILE < a.Nk) {
        stK.next(kt + TILE, tid);
        stV.next(kt + TILE, tid);
        stK.fetch();
        stV.fetch();
      }
    }
    f32x4_t s[4][2], dp[4][2];
    zero_acc(s);
    zero_acc(dp);
    s_product<DH>(s, tK, qf, lane);
    s_product<DH>(dp, tV, dof, lane);
    // (the same arithmetic on pairs -- v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32 -- measured SLOWER here and in dK/dV at d = 40:
    // 433 -> 447 us backward at 4 x 8 x 4096^2 x 40; the pair registers cost more moves than the packed issue saves)
#pragma unroll
    for (int sf = 0; sf < 4; ++sf)
#pragma unroll
      for (int of = 0; of < 2; ++of)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float p = LFOLD ? __builtin_amdgcn_exp2f(s[sf][of][e]) : __builtin_amdgcn_exp2f(__builtin_fmaf(s[sf][of][e], c, -lse2[of]));
          s[sf][of][e] = DFOLD ? p * dp[sf][of][e] : p * (dp[sf][of][e] - dl[of]);
        }
    if (kt + TILE > a.Nk) {  // last, partial tile: its padding rows repeat the last key row -> drop them
#pragma unroll
      for (int sf = 0; sf < 4; ++sf)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (kt + sf * 16 + (lane >> 4) * 4 + e >= a.Nk) s[sf][0][e] = s[sf][1][e] = 0.f;
    }
    bf16x8_t pb[2][2];
    pack_p(pb, s);
    t_product<DH, DV>(dq, tK, pb, lane);
    if constexpr (DMA) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      cur ^= 1;
    }
  }
  const float mq[2] = {a.scale, a.scale};
  const long ld_dq = a.ldgq > 0 ? a.ldgq : (long)a.H * a.d;  // dQ is written dense ([B][Nq][H*d]) whatever the row stride of q (q may be a column view), unless ldg says otherwise
  store_t<DV>(dq, a.dq + (long)b * a.Nq * ld_dq + h * a.d, ld_dq, q0, a.Nq, a.d, mq, lane);
}

// ------------------------------------------------------------------------------------------------ dK, dV
// DFOLD: as in attn_dq_kernel, mirrored -- here the dO rows are the streamed side: (-delta_hi, -delta_lo) of a row go into the two
// padding columns of its row in the dO tile (written with the tile's row statistics), the owner V fragments carry 1.0 there.
// LFOLD: mirrored as well -- (-lse2_hi, -lse2_lo) of a row go into the padding columns of its row in the Q tile, the owner K fragments carry 1.0.
template <int DH, int DV, bool DFOLD = false, bool LFOLD = false>
__global__ __launch_bounds__(256, (DH <= 64 ? 2 : 1)) void attn_dkv_kernel(const AttnArgs a) {
  constexpr bool DMA = DH > AQL_ATTN_DMA_ABOVE;   // see attn_fwd_kernel / DmaTile
  constexpr int IMG = TILE * RowPitch<DH>::value, NIMG = DMA ? 2 : 1;
  __shared__ __attribute__((aligned(1024))) char sQ[NIMG * IMG];
  __shared__ __attribute__((aligned(1024))) char sdO[NIMG * IMG];
  __shared__ __attribute__((aligned(16))) float sLse[NIMG * TILE];
  __shared__ __attribute__((aligned(16))) float sDelta[NIMG * TILE];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bx, h, bz;
  attn_block(bx, h, bz);
  const int b = bz / a.qsplit, split = bz - b * a.qsplit;
  const int k0 = bx * (4 * OWN) + wave * OWN;
  // cross-attention has 77 key rows: one owner workgroup per (b, h) would stream all of Q on 32 CUs.  Split the Q range
  // over qsplit workgroups instead; each writes an fp32 partial that a small kernel reduces (deterministic, no atomics)
  const int tiles_per_split = ((a.Nq + TILE - 1) / TILE + a.qsplit - 1) / a.qsplit;
  const int qt_lo = split * tiles_per_split * TILE;
  const int qt_hi = min(a.Nq, qt_lo + tiles_per_split * TILE);
  const bf16_t* qp = a.q + (long)b * a.Nq * a.ldq + h * a.d;
  const bf16_t* dop = a.dout + (long)b * a.Nq * a.ldo + h * a.d;
  const bf16_t* kp = a.k + (long)b * a.Nk * a.ldk + h * a.d;
  const bf16_t* vp = a.v + (long)b * a.Nk * a.ldv + h * a.d;
  bf16x8_t kf[2][DH / 32], vf[2][DH / 32];
  load_owner<DH>(kf, kp, a.ldk, k0, a.Nk, a.d, lane);
  load_owner<DH>(vf, vp, a.ldv, k0, a.Nk, a.d, lane);
  if constexpr (DFOLD) {
#pragma unroll
    for (int of = 0; of < 2; ++of)
#pragma unroll
      for (int ks = 0; ks < DH / 32; ++ks)
        if (ks == (a.d >> 5) && (lane >> 4) == ((a.d & 31) >> 3)) {
          uint4 v = *reinterpret_cast<uint4*>(&vf[of][ks]);
          v.x = 0x3F803F80u;   // columns d, d + 1 = 1.0
          vf[of][ks] = *reinterpret_cast<bf16x8_t*>(&v);
        }
  }
  if constexpr (LFOLD) {
#pragma unroll
    for (int of = 0; of < 2; ++of)
#pragma unroll
      for (int ks = 0; ks < DH / 32; ++ks)
        if (ks == (a.d >> 5) && (lane >> 4) == ((a.d & 31) >> 3)) {
          uint4 v = *reinterpret_cast<uint4*>(&kf[of][ks]);
          v.x = 0x3F803F80u;
          kf[of][ks] = *reinterpret_cast<bf16x8_t*>(&v);
        }
  }
  f32x4_t dk[DV / 16][2], dv[DV / 16][2];
  zero_acc(dk);
  zero_acc(dv);
  const float c = a.cexp;
  Stager<DH> stQ, stO;
  DmaTile<DH> dmQ, dmO;
  const float* lse_row = a.lse + ((long)b * a.H + h) * a.Nq;
  const float* delta_row = a.delta + ((long)b * a.H + h) * a.Nq;
  float lse_r = 0.f, delta_r = 0.f;   // threads 0..63: row statistics of the tile in flight
  auto fetch_stats = [&](int qt) {
    if (tid < TILE) {
      const int r = min(qt + tid, a.Nq - 1);
      lse_r = lse_row[r];
      delta_r = delta_row[r];
    }
  };
  auto put_stats = [&](int img, int qt) {
    if (tid < TILE) {
      const bool ok = (qt + tid) < a.Nq;   // rows past the end: p = exp2(s - inf) = 0
      sLse[img * TILE + tid] = ok ? lse_r * LOG2E : INFINITY;
      if constexpr (LFOLD) {
        const float l2 = ok ? lse_r * LOG2E : 1e30f;
        const uint32_t hi = pack_bf16x2(l2, 0.f) & 0xffffu;
        const uint32_t lo = pack_bf16x2(l2 - bf16lo(hi), 0.f) & 0xffffu;
        *reinterpret_cast<uint32_t*>(sQ + img * IMG + tile_off<DH>(tid, a.d >> 3)) = (hi ^ 0x8000u) | ((lo ^ 0x8000u) << 16);
      }
      if constexpr (DFOLD) {
        const uint32_t hi = pack_bf16x2(delta_r, 0.f) & 0xffffu;
        const uint32_t lo = pack_bf16x2(delta_r - bf16lo(hi), 0.f) & 0xffffu;
        *reinterpret_cast<uint32_t*>(sdO + img * IMG + tile_off<DH>(tid, a.d >> 3)) = ok ? ((hi ^ 0x8000u) | ((lo ^ 0x8000u) << 16)) : 0u;
      } else {
        sDelta[img * TILE + tid] = ok ? delta_r : 0.f;
      }
    }
  };
  if constexpr (DMA) {
    dmQ.init(qp, a.ldq, a.Nq, a.d, wave, lane);
    dmO.init(dop, a.ldo, a.Nq, a.d, wave, lane);
    DmaTile<DH>::pad(sQ, 2, a.d, tid);
    DmaTile<DH>::pad(sdO, 2, a.d, tid);
    if constexpr (DFOLD || LFOLD) __syncthreads();   // put_stats writes into padding chunks that other threads have just zeroed
    if (qt_lo < qt_hi) {
      dmQ.issue(sQ, qt_lo);
      dmO.issue(sdO, qt_lo);
      fetch_stats(qt_lo);
      put_stats(0, qt_lo);
      if (qt_lo + TILE < qt_hi) fetch_stats(qt_lo + TILE);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  } else {
    stQ.init(sQ, qp, a.ldq, qt_lo, a.Nq, a.d, tid);
    stO.init(sdO, dop, a.ldo, qt_lo, a.Nq, a.d, tid);
    if (qt_lo < qt_hi) {
      stQ.fetch();
      stO.fetch();
      fetch_stats(qt_lo);
    }
  }
  int cur = 0;
  for (int qt = qt_lo; qt < qt_hi; qt += TILE) {
    if constexpr (DMA) {
      if (qt + TILE < qt_hi) {   // tile t+1: operands by LDS-DMA into the other image, its row statistics from the registers
        dmQ.issue(sQ + (cur ^ 1) * IMG, qt + TILE);
        dmO.issue(sdO + (cur ^ 1) * IMG, qt + TILE);
        put_stats(cur ^ 1, qt + TILE);
        if (qt + 2 * TILE < qt_hi) fetch_stats(qt + 2 * TILE);
      }
    } else {
      __syncthreads();
      stQ.commit(sQ);
      stO.commit(sdO);
      put_stats(0, qt);
      __syncthreads();
      if (qt + TILE < qt_hi) {
        stQ.next(qt + TILE, tid);
        stO.next(qt + TILE, tid);
        stQ.fetch();
        stO.fetch();
        fetch_stats(qt + TILE);
      }
    }
    const char* tQ = sQ + cur * IMG;
    const char* tO = sdO + cur * IMG;
    f32x4_t s[4][2], dp[4][2];
    zero_acc(s);
    zero_acc(dp);
    s_product<DH>(s, tQ, kf, lane);
    s_product<DH>(dp, tO, vf, lane);
    f32x4_t ds[4][2];
#pragma unroll
    for (int sf = 0; sf < 4; ++sf) {
      const float4 ls = *reinterpret_cast<const float4*>(&sLse[cur * TILE + sf * 16 + (lane >> 4) * 4]);
      const float4 de = DFOLD ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4*>(&sDelta[cur * TILE + sf * 16 + (lane >> 4) * 4]);
      const float lsv[4] = {ls.x, ls.y, ls.z, ls.w};
      const float dev[4] = {de.x, de.y, de.z, de.w};
#pragma unroll
      for (int of = 0; of < 2; ++of)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float p = LFOLD ? __builtin_amdgcn_exp2f(s[sf][of][e]) : __builtin_amdgcn_exp2f(__builtin_fmaf(s[sf][of][e], c, -lsv[e]));
          s[sf][of][e] = p;
          ds[sf][of][e] = DFOLD ? p * dp[sf][of][e] : p * (dp[sf][of][e] - dev[e]);
        }
    }
    bf16x8_t pb[2][2];
    pack_p(pb, s);
    t_product<DH, DV>(dv, tO, pb, lane);
    pack_p(pb, ds);
    t_product<DH, DV>(dk, tQ, pb, lane);
    if constexpr (DMA) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      cur ^= 1;
    }
  }
  const long ldo_kv = a.ldgkv > 0 ? a.ldgkv : (long)a.H * a.d;
  if (a.qsplit > 1) {
    const long slab = (long)a.qsplit * a.B * a.H * a.Nk * a.d;
    float* pk = a.part + (((long)split * a.B + b) * a.H + h) * a.Nk * a.d;
    store_t_f32<DV>(dk, pk, a.d, k0, a.Nk, a.d, lane);
    store_t_f32<DV>(dv, pk + slab, a.d, k0, a.Nk, a.d, lane);
    return;
  }
  const float mk[2] = {a.cnat, a.cnat}, mv[2] = {1.f, 1.f};
  store_t<DV>(dk, a.dk + (long)b * a.Nk * ldo_kv + h * a.d, ldo_kv, k0, a.Nk, a.d, mk, lane);
  store_t<DV>(dv, a.dv + (long)b * a.Nk * ldo_kv + h * a.d, ldo_kv, k0, a.Nk, a.d, mv, lane);
}

// dk/dv[b][row][h*d + c] = bf16(mul * sum_split part[which][split][b][h][row][c]); 4 columns per thread
__global__ __launch_bounds__(256) void attn_dkv_reduce_kernel(const AttnArgs a) {
  const int d4 = a.d >> 2;
  const long per = (long)a.B * a.H * a.Nk * d4, total = 2 * per;
  const long slab = (long)a.qsplit * a.B * a.H * a.Nk * a.d;
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long)gridDim.x * blockDim.x) {
    const int which = id >= per;
    long r = id - which * per;
    const int c = (int)(r % d4) * 4;
    r /= d4;
    const int row = (int)(r % a.Nk);
    r /= a.Nk;
    const int h = (int)(r % a.H), b = (int)(r / a.H);
    const float* p = a.part + which * slab + (((long)b * a.H + h) * a.Nk + row) * a.d + c;
    const long sstride = (long)a.B * a.H * a.Nk * a.d;
    float4 acc = *reinterpret_cast<const float4*>(p);
    for (int sp = 1; sp < a.qsplit; ++sp) {
      const float4 t = *reinterpret_cast<const float4*>(p + sp * sstride);
      acc.x += t.x, acc.y += t.y, acc.z += t.z, acc.w += t.w;
    }
    const float mul = which ? 1.f : a.cnat;
    bf16_t* out = (which ? a.dv : a.dk) + ((long)b * a.Nk + row) * (a.ldgkv > 0 ? a.ldgkv : (long)a.H * a.d) + h * a.d + c;
    *reinterpret_cast<uint2*>(out) = make_uint2(pack_bf16x2(acc.x * mul, acc.y * mul), pack_bf16x2(acc.z * mul, acc.w * mul));
  }
}

// ------------------------------------------------------------------------------------------------ short key side (cross-attention)
// Text-state attention has Nk = 77 keys (original_unet.py:688-704 with encoder_hidden_states): negligible FLOPs, and in the streaming
// kernels above every workgroup staged the same two K/V tiles behind four barriers, handled one 256-row Q block and retired -- 21 us per
// launch at the 64x64 level for 42 MB of Q / O traffic, 14 us at 16x16 for 2.6 MB (profiles/r03_step_sequence.txt).  Here the whole key
// side (<= CTX_ROWS rows) is RESIDENT: K and V are staged once into two LDS images, ONE barrier, and then the four wavefronts run free,
// each walking 32-row owner blocks of the workgroup's Q range with the next block's Q (and dO / O) fragments already in flight.  The
// softmax is the plain two-pass one (all 80 scores of a row are in registers: no running maximum, no rescale of the accumulator).
constexpr int CTX_ROWS = 80;   // five 16-row fragments
constexpr int CTX_NSF = CTX_ROWS / 16;

template <int DH>
__device__ __forceinline__ void ctx_stage(char* img, const bf16_t* g, long ld, int nrows, int d, int tid) {
  constexpr int CPR = DH / 8;
  constexpr int NIT = (CTX_ROWS * CPR + 255) / 256;
  uint4 v[NIT];
  // every load first (unconditional, clamped address, masked after): one round trip for the whole image instead of NIT
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int id = tid + it * 256;
    const int row = id / CPR, c = id - row * CPR;
    const bool ok = (id < CTX_ROWS * CPR) & (row < nrows) & (c * 8 < d);
    const uint4 x = *reinterpret_cast<const uint4*>(g + (ok ? (long)row * ld + c * 8 : 0));
    v[it] = mask4(x, ok);   // rows >= nrows and the d..DH padding: zero
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int id = tid + it * 256;
    const int row = id / CPR, c = id - row * CPR;
    if (id < CTX_ROWS * CPR) *reinterpret_cast<uint4*>(img + tile_off<DH>(row, c)) = v[it];
  }
}

// acc[sf][of] = image rows sf*16.. (A) x owner frag of (B), all CTX_NSF fragments of the resident image
template <int DH>
__device__ __forceinline__ void ctx_s_product(f32x4_t (&acc)[CTX_NSF][2], const char* img, const bf16x8_t (&own)[2][DH / 32], int lane) {
#pragma unroll
  for (int s = 0; s < DH / 32; ++s) {
    bf16x8_t a[CTX_NSF];
#pragma unroll
    for (int sf = 0; sf < CTX_NSF; ++sf)
      a[sf] = *reinterpret_cast<const bf16x8_t*>(img + tile_off<DH>(sf * 16 + (lane & 15), s * 4 + (lane >> 4)));
#pragma unroll
    for (int sf = 0; sf < CTX_NSF; ++sf)
#pragma unroll
      for (int of = 0; of < 2; ++of) {
        if (s == 0) acc[sf][of] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[sf], own[of][s], f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        else acc[sf][of] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[sf], own[of][s], acc[sf][of], 0, 0, 0);
      }
  }
}

// p (fp32, rows = key sf*16 + g*4 + reg) -> B fragments of the three 32-key steps; the second half of step 2 (keys 80..95) is zero
__device__ __forceinline__ void ctx_pack_p(bf16x8_t (&pb)[3][2], const f32x4_t (&p)[CTX_NSF][2]) {
#pragma unroll
  for (int s2 = 0; s2 < 3; ++s2)
#pragma unroll
    for (int of = 0; of < 2; ++of) {
      uint4 v;
      v.x = pack_bf16x2(p[2 * s2][of][0], p[2 * s2][of][1]);
      v.y = pack_bf16x2(p[2 * s2][of][2], p[2 * s2][of][3]);
      if (2 * s2 + 1 < CTX_NSF) {
        v.z = pack_bf16x2(p[(2 * s2 + 1) % CTX_NSF][of][0], p[(2 * s2 + 1) % CTX_NSF][of][1]);
        v.w = pack_bf16x2(p[(2 * s2 + 1) % CTX_NSF][of][2], p[(2 * s2 + 1) % CTX_NSF][of][3]);
      } else {
        v.z = 0u;
        v.w = 0u;
      }
      pb[s2][of] = *reinterpret_cast<bf16x8_t*>(&v);
    }
}

// acc[df][of] += (image)^T frag df (A) x pb (B) over the 80 resident rows (see t_product; the missing rows 80..95 of step 2 re-read
// rows 64..79 -- finite -- against zero probabilities)
template <int DH, int DV>
__device__ __forceinline__ void ctx_t_product(f32x4_t (&acc)[DV / 16][2], const char* img, const bf16x8_t (&pb)[3][2], int lane) {
  const int p = lane & 15, g = lane >> 4;
#pragma unroll
  for (int s2 = 0; s2 < 3; ++s2) {
    const int row = s2 * 32 + g * 4 + (p >> 2);
#pragma unroll
    for (int df = 0; df < DV / 16; ++df) {
      const char* base = img + tile_off<DH>(row, df * 2 + ((p & 3) >> 1)) + (p & 1) * 8;
      const v4s_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s_t __attribute__((address_space(3)))*)(base));
      const v4s_t hi = s2 < 2 ? __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                                    (v4s_t __attribute__((address_space(3)))*)(base + 16 * RowPitch<DH>::value))
                              : lo;
      const bf16x8_t a = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
      for (int of = 0; of < 2; ++of)
        acc[df][of] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, pb[s2][of], acc[df][of], 0, 0, 0);
    }
  }
}

// grid (ceil(Nq / (128 * a.qsplit)), H, B): a.qsplit = owner blocks per wavefront (the forward / dQ kernels reuse the field), <= NB.
// The Q fragments of ALL the wavefront's blocks are requested before the barrier (16 registers per block at d = 40): with one block
// of look-ahead a wavefront had 2.5 KB in flight, 20 KB per CU, and the launch ran at the latency-bound 2.2 TB/s.
template <int DH>
struct CtxNB {
  static constexpr int fwd = DH <= 64 ? 4 : (DH <= 96 ? 2 : 1);
  static constexpr int bwd = DH <= 64 ? 2 : 1;
};
template <int DH, int DV>
__global__ __launch_bounds__(256, (DH <= 96 ? 2 : 1)) void attn_ctx_fwd_kernel(const AttnArgs a) {
  constexpr int IMG = CTX_ROWS * RowPitch<DH>::value;
  constexpr int NB = CtxNB<DH>::fwd;
  __shared__ __attribute__((aligned(1024))) char sK[IMG];
  __shared__ __attribute__((aligned(1024))) char sV[IMG];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bx, h, b;
  attn_block(bx, h, b);
  const int nb = a.qsplit;
  const int wg_q0 = bx * (4 * OWN * nb);
  const bf16_t* qp = a.q + (long)b * a.Nq * a.ldq + h * a.d;
  bf16x8_t qf[NB][2][DH / 32];
#pragma unroll
  for (int i = 0; i < NB; ++i)
    if (i < nb) load_owner<DH>(qf[i], qp, a.ldq, wg_q0 + (i * 4 + wave) * OWN, a.Nq, a.d, lane);
  ctx_stage<DH>(sK, a.k + (long)b * a.Nk * a.ldk + h * a.d, a.ldk, a.Nk, a.d, tid);
  ctx_stage<DH>(sV, a.v + (long)b * a.Nk * a.ldv + h * a.d, a.ldv, a.Nk, a.d, tid);
  __syncthreads();
  const float c = a.cexp;
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int q0 = wg_q0 + (i * 4 + wave) * OWN;
    if (i >= nb || q0 >= a.Nq) break;   // wave-uniform
    f32x4_t s[CTX_NSF][2];
    ctx_s_product<DH>(s, sK, qf[i], lane);
    float inv[2], lsev[2];
#pragma unroll
    for (int of = 0; of < 2; ++of) {
      float mx = -INFINITY;
#pragma unroll
      for (int sf = 0; sf < CTX_NSF; ++sf)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (sf * 16 + (lane >> 4) * 4 + e >= a.Nk) s[sf][of][e] = -INFINITY;
          mx = fmaxf(mx, s[sf][of][e]);
        }
      mx = group4_max(mx);
      const float mxc = mx * c;
      float rs = 0.f;
#pragma unroll
      for (int sf = 0; sf < CTX_NSF; ++sf)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[sf][of][e], c, -mxc));
          s[sf][of][e] = p;
          rs += p;
        }
      rs = group4_sum(rs);
      inv[of] = 1.f / rs;
      lsev[of] = mx * a.cnat + logf(rs);
    }
    bf16x8_t pb[3][2];
    ctx_pack_p(pb, s);
    f32x4_t o[DV / 16][2];
    zero_acc(o);
    ctx_t_product<DH, DV>(o, sV, pb, lane);
    store_t<DV>(o, a.out + (long)b * a.Nq * a.ldo + h * a.d, a.ldo, q0, a.Nq, a.d, inv, lane);
    if ((lane >> 4) == 0) {
#pragma unroll
      for (int of = 0; of < 2; ++of) {
        const int row = q0 + of * 16 + (lane & 15);
        if (row < a.Nq) a.lse[((long)b * a.H + h) * a.Nq + row] = lsev[of];
      }
    }
  }
}

// dQ (and delta) with the key side resident: owner = Q, dO (and O for delta)
template <int DH, int DV>
__global__ __launch_bounds__(256, (DH <= 96 ? 2 : 1)) void attn_ctx_dq_kernel(const AttnArgs a) {
  constexpr int IMG = CTX_ROWS * RowPitch<DH>::value;
  __shared__ __attribute__((aligned(1024))) char sK[IMG];
  __shared__ __attribute__((aligned(1024))) char sV[IMG];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bx, h, b;
  attn_block(bx, h, b);
  const int nb = a.qsplit;
  const int wg_q0 = bx * (4 * OWN * nb);
  const bf16_t* qp = a.q + (long)b * a.Nq * a.ldq + h * a.d;
  const bf16_t* dop = a.dout + (long)b * a.Nq * a.ldo + h * a.d;
  const bf16_t* op = a.o + (long)b * a.Nq * a.ldo + h * a.d;
  constexpr int NB = CtxNB<DH>::bwd;
  bf16x8_t qfa[NB][2][DH / 32], dofa[NB][2][DH / 32], ofa[NB][2][DH / 32];
#pragma unroll
  for (int i = 0; i < NB; ++i)
    if (i < nb) {   // every operand of every block of this wavefront is in flight before the barrier
      const int q0 = wg_q0 + (i * 4 + wave) * OWN;
      load_owner<DH>(qfa[i], qp, a.ldq, q0, a.Nq, a.d, lane);
      load_owner<DH>(dofa[i], dop, a.ldo, q0, a.Nq, a.d, lane);
      load_owner<DH>(ofa[i], op, a.ldo, q0, a.Nq, a.d, lane);
    }
  ctx_stage<DH>(sK, a.k + (long)b * a.Nk * a.ldk + h * a.d, a.ldk, a.Nk, a.d, tid);
  ctx_stage<DH>(sV, a.v + (long)b * a.Nk * a.ldv + h * a.d, a.ldv, a.Nk, a.d, tid);
  __syncthreads();
  const float c = a.cexp;
  const long ld_dq = a.ldgq > 0 ? a.ldgq : (long)a.H * a.d;
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int q0 = wg_q0 + (i * 4 + wave) * OWN;
    if (i >= nb || q0 >= a.Nq) break;   // wave-uniform
    const bf16x8_t (&qf)[2][DH / 32] = qfa[i];
    const bf16x8_t (&dof)[2][DH / 32] = dofa[i];
    float lse2[2], dl[2];
    {
      const bf16x8_t (&ofr)[2][DH / 32] = ofa[i];
#pragma unroll
      for (int of = 0; of < 2; ++of) {
        float acc = 0.f;
#pragma unroll
        for (int x = 0; x < DH / 32; ++x) {
          const uint4 u = *reinterpret_cast<const uint4*>(&dof[of][x]);
          const uint4 y = *reinterpret_cast<const uint4*>(&ofr[of][x]);
          acc += bf16lo(u.x) * bf16lo(y.x) + bf16hi(u.x) * bf16hi(y.x) + bf16lo(u.y) * bf16lo(y.y) + bf16hi(u.y) * bf16hi(y.y) +
                 bf16lo(u.z) * bf16lo(y.z) + bf16hi(u.z) * bf16hi(y.z) + bf16lo(u.w) * bf16lo(y.w) + bf16hi(u.w) * bf16hi(y.w);
        }
        dl[of] = group4_sum(acc);
      }
    }
#pragma unroll
    for (int of = 0; of < 2; ++of) {
      const int row = q0 + of * 16 + (lane & 15);
      const bool ok = row < a.Nq;
      lse2[of] = ok ? a.lse[((long)b * a.H + h) * a.Nq + row] * LOG2E : INFINITY;
      if (ok && (lane >> 4) == 0) a.delta[((long)b * a.H + h) * a.Nq + row] = dl[of];
    }
    f32x4_t s[CTX_NSF][2], dp[CTX_NSF][2];
    ctx_s_product<DH>(s, sK, qf, lane);
    ctx_s_product<DH>(dp, sV, dof, lane);
#pragma unroll
    for (int sf = 0; sf < CTX_NSF; ++sf)
#pragma unroll
      for (int of = 0; of < 2; ++of)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[sf][of][e], c, -lse2[of]));
          const bool live = sf * 16 + (lane >> 4) * 4 + e < a.Nk;   // zero key rows past Nk score 0, not -inf: drop them
          s[sf][of][e] = live ? p * (dp[sf][of][e] - dl[of]) : 0.f;
        }
    bf16x8_t pb[3][2];
    ctx_pack_p(pb, s);
    f32x4_t dq[DV / 16][2];
    zero_acc(dq);
    ctx_t_product<DH, DV>(dq, sK, pb, lane);
    const float mq[2] = {a.scale, a.scale};
    store_t<DV>(dq, a.dq + (long)b * a.Nq * ld_dq + h * a.d, ld_dq, q0, a.Nq, a.d, mq, lane);
  }
}

// owner blocks per wavefront: enough workgroups for two per CU, at most `nbmax` (the kernel's register budget) blocks
inline int ctx_blocks(const AttnArgs& a, int nbmax) {
  static const int force = AQL_TUNE_INT("AQL_ATTN_CTX_NB", 0);   // tuning hook
  const long wgs1 = (long)aql_cdiv(a.Nq, 4 * OWN) * a.H * a.B;
  int nb = force > 0 ? force : (int)(wgs1 / 512);
  return nb < 1 ? 1 : (nb > nbmax ? nbmax : nb);
}
inline bool ctx_on(const AttnArgs& a) {
  static const int en = AQL_TUNE_INT("AQL_ATTN_CTX", 1);   // A/B hook: 0 = the streaming kernels
  return en && a.Nk <= CTX_ROWS;
}

template <int DH, int DV>
int launch_fwd(const AttnArgs& a0, hipStream_t st) {
  if (ctx_on(a0)) {
    AttnArgs a = a0;
    a.qsplit = ctx_blocks(a, CtxNB<DH>::fwd);
    hipLaunchKernelGGL((attn_ctx_fwd_kernel<DH, DV>), dim3(aql_cdiv(a.Nq, 4 * OWN * a.qsplit), a.H, a.B), dim3(256), 0, st, a);
    return 0;
  }
  const AttnArgs& a = a0;
  static const int force = AQL_TUNE_INT("AQL_ATTN_NOF", 0);  // tuning hook
  static const int ones = AQL_TUNE_INT("AQL_ATTN_ONES", 1);  // tuning hook
  static const int fold = getenv("AQL_ATTN_FOLD") ? atoi(getenv("AQL_ATTN_FOLD")) : 1;  // A/B hook: 0 = the running-maximum loop, 2 = the shift in the MFMA
  if constexpr (DH <= 96) {
    if (fold && ones && a.d < DV) {   // the denominator rides in the P.V product: the loop needs no row statistics after its first tile
      const bool big = DH <= 64 && (force == 4 || (force == 0 && a.Nq >= 2048 && (long)aql_cdiv(a.Nq, 256) * a.H * a.B >= 512));
      const dim3 g4(aql_cdiv(a.Nq, 256), a.H, a.B), g2(aql_cdiv(a.Nq, 128), a.H, a.B);
      // q pre-multiplied by scale log2(e) (aql_sdpa_fwd_qpre): the loop whose shift rides in the S-product, at the default loop's precision
      if constexpr (DH <= 64 && DV <= 48) {
        if (a.qpre && a.d < DH) {
          if (big) hipLaunchKernelGGL((attn_fwd_kernel<DH, DV, 4, true, 2>), g4, dim3(256), 0, st, a);
          else hipLaunchKernelGGL((attn_fwd_kernel<DH, DV, 2, true, 2>), g2, dim3(256), 0, st, a);
          return 0;
        }
      }
#ifdef AQL_EXPERIMENTS   // AQL_ATTN_FOLD=2 (the shift inside the S-product: 210 vs 231 us, twice the rounding error; profiles/r04_attention_fold.txt)
      if (fold == 2 && a.d < DH) {
        if constexpr (DH <= 64) {
          if (big) {
            hipLaunchKernelGGL((attn_fwd_kernel<DH, DV, 4, true, 2>), g4, dim3(256), 0, st, a);
            return 0;
          }
        }
        hipLaunchKernelGGL((attn_fwd_kernel<DH, DV, 2, true, 2>), g2, dim3(256), 0, st, a);
        return 0;
      }
#endif
      if constexpr (DH <= 64 && DV <= 48) {   // (64 rows per wavefront at DV = 64 would spill two registers: no head of the U-Net needs it)
        if (big) {
          hipLaunchKernelGGL((attn_fwd_kernel<DH, DV, 4, true, 1>), g4, dim3(256), 0, st, a);
          return 0;
        }
      }
      hipLaunchKernelGGL((attn_fwd_kernel<DH, DV, 2, true, 1>), g2, dim3(256), 0, st, a);
      return 0;
    }
  }
  if constexpr (DH <= 64) {
    // 64 rows per wavefront only while that still gives two workgroups per CU (one guided image = 2 x 8 heads x 16 row blocks = 256
    // workgroups of 256 rows: 97 us, against 93.5 us as 512 workgroups of 128 rows; a single sample 89 vs 62 us)
    if (force == 4 || (force == 0 && a.Nq >= 2048 && (long)aql_cdiv(a.Nq, 256) * a.H * a.B >= 512)) {
      if (a.d < DV && ones) hipLaunchKernelGGL((attn_fwd_kernel<DH, DV, 4, true>), dim3(aql_cdiv(a.Nq, 256), a.H, a.B), dim3(256), 0, st, a);
      else hipLaunchKernelGGL((attn_fwd_kernel<DH, DV, 4, false>), dim3(aql_cdiv(a.Nq, 256), a.H, a.B), dim3(256), 0, st, a);
      return 0;
    }
  }
  if constexpr (DH <= 96) {   // d = 40 / 80 leave padding columns in the V tile: the denominator rides in the P.V product
    if (a.d < DV && ones) {
      hipLaunchKernelGGL((attn_fwd_kernel<DH, DV, 2, true>), dim3(aql_cdiv(a.Nq, 128), a.H, a.B), dim3(256), 0, st, a);
      return 0;
    }
  }
  hipLaunchKernelGGL((attn_fwd_kernel<DH, DV, 2, false>), dim3(aql_cdiv(a.Nq, 128), a.H, a.B), dim3(256), 0, st, a);
  return 0;
}
template <int DH, int DV>
int launch_bwd(const AttnArgs& a, hipStream_t st) {
  static const int dfold_on = getenv("AQL_ATTN_DFOLD") ? atoi(getenv("AQL_ATTN_DFOLD")) : 1;   // A/B hook: 0 = subtract delta per element
  const bool dfold = DH <= 96 && dfold_on && a.d < DH;   // two spare head columns (d % 8 == 0)
  static const int lfold_on = getenv("AQL_ATTN_LFOLD") ? atoi(getenv("AQL_ATTN_LFOLD")) : 1;   // A/B hook: 0 = exp2(fma(s, 1, -lse)) on a pre-scaled q
  const bool lfold = dfold && lfold_on && a.qpre && DH <= 64 && DV <= 48 && !ctx_on(a);
  if (ctx_on(a)) {
    AttnArgs c = a;
    c.qsplit = ctx_blocks(a, CtxNB<DH>::bwd);
    hipLaunchKernelGGL((attn_ctx_dq_kernel<DH, DV>), dim3(aql_cdiv(a.Nq, 4 * OWN * c.qsplit), a.H, a.B), dim3(256), 0, st, c);
  } else if (dfold) {
    if constexpr (DH <= 64 && DV <= 48) {
      if (lfold) hipLaunchKernelGGL((attn_dq_kernel<DH, DV, true, true>), dim3(aql_cdiv(a.Nq, 4 * OWN), a.H, a.B), dim3(256), 0, st, a);
    }
    if constexpr (DH <= 96) {
      if (!lfold) hipLaunchKernelGGL((attn_dq_kernel<DH, DV, true>), dim3(aql_cdiv(a.Nq, 4 * OWN), a.H, a.B), dim3(256), 0, st, a);
    }
  } else {
    hipLaunchKernelGGL((attn_dq_kernel<DH, DV>), dim3(aql_cdiv(a.Nq, 4 * OWN), a.H, a.B), dim3(256), 0, st, a);
  }
  if (dfold) {
    if constexpr (DH <= 64 && DV <= 48) {
      if (lfold) hipLaunchKernelGGL((attn_dkv_kernel<DH, DV, true, true>), dim3(aql_cdiv(a.Nk, 4 * OWN), a.H, a.B * a.qsplit), dim3(256), 0, st, a);
    }
    if constexpr (DH <= 96) {
      if (!lfold) hipLaunchKernelGGL((attn_dkv_kernel<DH, DV, true>), dim3(aql_cdiv(a.Nk, 4 * OWN), a.H, a.B * a.qsplit), dim3(256), 0, st, a);
    }
  } else {
    hipLaunchKernelGGL((attn_dkv_kernel<DH, DV>), dim3(aql_cdiv(a.Nk, 4 * OWN), a.H, a.B * a.qsplit), dim3(256), 0, st, a);
  }
  if (a.qsplit > 1) {
    const long n = 2L * a.B * a.H * a.Nk * (a.d / 4);
    hipLaunchKernelGGL(attn_dkv_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a);
  }
  return 0;
}

}  // namespace

static int sdpa_fwd_impl(int qpre, const bf16_t* q, long ldq, const bf16_t* k, long ldk, const bf16_t* v, long ldv, int B,
                         int H, int Nq, int Nk, int d, float scale, bf16_t* o, long ldo, float* lse,
                         hipStream_t stream) {
  AQL_CHECK_ARG(q && k && v && o && lse, "aql_sdpa_fwd: null operand");
  AQL_CHECK_ARG(d % 8 == 0 && d <= 160 && ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 8 == 0 && Nk > 0,
                "aql_sdpa_fwd: unsupported head dim %d or strides", d);
  AttnArgs a{};
  a.q = q; a.k = k; a.v = v; a.out = o; a.lse = lse;
  a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo;
  a.B = B; a.H = H; a.Nq = Nq; a.Nk = Nk; a.d = d; a.scale = scale;
  a.qpre = qpre; a.cexp = qpre ? 1.f : scale * LOG2E; a.cnat = qpre ? 0.6931471805599453f : scale;
  if (d <= 48) launch_fwd<64, 48>(a, stream);
  else if (d <= 64) launch_fwd<64, 64>(a, stream);
  else if (d <= 96) launch_fwd<96, 96>(a, stream);
  else if (d <= 128) launch_fwd<128, 128>(a, stream);
  else launch_fwd<160, 160>(a, stream);
  AQL_CHECK_LAUNCH("aql_sdpa_fwd");
  return AQL_OK;
}

extern "C" int aql_sdpa_fwd(const bf16_t* q, long ldq, const bf16_t* k, long ldk, const bf16_t* v, long ldv, int B,
                            int H, int Nq, int Nk, int d, float scale, bf16_t* o, long ldo, float* lse,
                            hipStream_t stream) {
  return sdpa_fwd_impl(0, q, ldq, k, ldk, v, ldv, B, H, Nq, Nk, d, scale, o, ldo, lse, stream);
}

// q is PRE-MULTIPLIED by scale * log2(e) (rounded to bf16 once, in the epilogue of the launch that produced it: aql_lora_chain_fwd's
// `oscale`); everything else as aql_sdpa_fwd, and lse is the same natural-log quantity.
extern "C" int aql_sdpa_fwd_qpre(const bf16_t* q, long ldq, const bf16_t* k, long ldk, const bf16_t* v, long ldv, int B,
                                 int H, int Nq, int Nk, int d, float scale, bf16_t* o, long ldo, float* lse,
                                 hipStream_t stream) {
  return sdpa_fwd_impl(1, q, ldq, k, ldk, v, ldv, B, H, Nq, Nk, d, scale, o, ldo, lse, stream);
}

static int sdpa_bwd_impl(int qpre, const bf16_t* q, long ldq, const bf16_t* k, long ldk, const bf16_t* v, long ldv,
                            const bf16_t* o, const bf16_t* dout, long ldo, const float* lse, float* delta, int B, int H,
                            int Nq, int Nk, int d, float scale, bf16_t* dq, bf16_t* dk, bf16_t* dv, float* ws,
                            size_t ws_bytes, hipStream_t stream, long ldgq = 0, long ldgkv = 0) {
  AQL_CHECK_ARG(q && k && v && o && dout && lse && delta && dq && dk && dv, "aql_sdpa_bwd: null operand");
  AQL_CHECK_ARG((ldgq == 0 || (ldgq % 8 == 0 && ldgq >= (long)H * d)) && (ldgkv == 0 || (ldgkv % 8 == 0 && ldgkv >= (long)H * d)),
                "aql_sdpa_bwd: gradient row strides %ld / %ld", ldgq, ldgkv);
  AQL_CHECK_ARG(d % 8 == 0 && d <= 160 && ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 8 == 0 && Nk > 0,
                "aql_sdpa_bwd: unsupported head dim %d or strides", d);
  AttnArgs a{};
  a.q = q; a.k = k; a.v = v; a.o = o; a.dout = dout; a.lse = const_cast<float*>(lse); a.delta = delta;
  a.dq = dq; a.dk = dk; a.dv = dv;
  a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo;
  a.ldgq = ldgq; a.ldgkv = ldgkv;
  a.B = B; a.H = H; a.Nq = Nq; a.Nk = Nk; a.d = d; a.scale = scale;
  a.qpre = qpre; a.cexp = qpre ? 1.f : scale * LOG2E; a.cnat = qpre ? 0.6931471805599453f : scale;
  // split the streamed Q range when the key side alone cannot fill the chip (cross-attention: Nk = 77)
  a.qsplit = 1;
  a.part = ws;
  const long owner_wgs = (long)aql_cdiv(Nk, 4 * OWN) * H * B;
  if (ws != nullptr && owner_wgs < 256 && Nq >= 4 * TILE) {
    int sp = (int)(512 / owner_wgs);
    const int qtiles = aql_cdiv(Nq, TILE);
    if (sp > qtiles / 2) sp = qtiles / 2;   // at least two Q tiles per workgroup
    while (sp > 1 && (size_t)2 * sp * B * H * Nk * d * sizeof(float) > ws_bytes) --sp;
    if (sp > 1) a.qsplit = sp;
  }
  if (d <= 48) launch_bwd<64, 48>(a, stream);
  else if (d <= 64) launch_bwd<64, 64>(a, stream);
  else if (d <= 96) launch_bwd<96, 96>(a, stream);
  else if (d <= 128) launch_bwd<128, 128>(a, stream);
  else launch_bwd<160, 160>(a, stream);
  AQL_CHECK_LAUNCH("aql_sdpa_bwd");
  return AQL_OK;
}

extern "C" int aql_sdpa_bwd(const bf16_t* q, long ldq, const bf16_t* k, long ldk, const bf16_t* v, long ldv,
                            const bf16_t* o, const bf16_t* dout, long ldo, const float* lse, float* delta, int B, int H,
                            int Nq, int Nk, int d, float scale, bf16_t* dq, bf16_t* dk, bf16_t* dv, float* ws,
                            size_t ws_bytes, hipStream_t stream) {
  return sdpa_bwd_impl(0, q, ldq, k, ldk, v, ldv, o, dout, ldo, lse, delta, B, H, Nq, Nk, d, scale, dq, dk, dv, ws, ws_bytes, stream);
}

// Backward of aql_sdpa_fwd_qpre: q is the pre-multiplied tensor the forward saw; dq is the gradient of the UNSCALED q (what the
// producing linear's backward expects), dk / dv as always.
extern "C" int aql_sdpa_bwd_qpre(const bf16_t* q, long ldq, const bf16_t* k, long ldk, const bf16_t* v, long ldv,
                                 const bf16_t* o, const bf16_t* dout, long ldo, const float* lse, float* delta, int B, int H,
                                 int Nq, int Nk, int d, float scale, bf16_t* dq, bf16_t* dk, bf16_t* dv, float* ws,
                                 size_t ws_bytes, hipStream_t stream) {
  return sdpa_bwd_impl(1, q, ldq, k, ldk, v, ldv, o, dout, ldo, lse, delta, B, H, Nq, Nk, d, scale, dq, dk, dv, ws, ws_bytes, stream);
}

// aql_sdpa_bwd / aql_sdpa_bwd_qpre (qpre = 0 / 1) whose gradients are written with row strides of ldg_q (dq) and ldg_kv (dk, dv)
// elements (0 = dense): dq | dk | dv as the column blocks of ONE [B][N][3 H d] buffer (self-attention), or dk | dv as the column blocks of
// one [B][Nk][2 H d] buffer (text-state attention) -- the activation-side operand of the grouped backward of the projections
// (aql_gemm_bf16_grouped), with no gather copy in between.
extern "C" int aql_sdpa_bwd_ex(int qpre, const bf16_t* q, long ldq, const bf16_t* k, long ldk, const bf16_t* v, long ldv,
                               const bf16_t* o, const bf16_t* dout, long ldo, const float* lse, float* delta, int B, int H, int Nq,
                               int Nk, int d, float scale, bf16_t* dq, bf16_t* dk, bf16_t* dv, long ldg_q, long ldg_kv, float* ws,
                               size_t ws_bytes, hipStream_t stream) {
  return sdpa_bwd_impl(qpre ? 1 : 0, q, ldq, k, ldk, v, ldv, o, dout, ldo, lse, delta, B, H, Nq, Nk, d, scale, dq, dk, dv, ws, ws_bytes,
                       stream, ldg_q, ldg_kv);
}
