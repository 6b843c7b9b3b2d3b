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

constexpr int BM = 128, CH = 320, NTH = 512, NKT = CH / 32, LR = 32, MAXS = 4;
constexpr int RES_BYTES = NKT * BM * 64;           // 81920
constexpr int W_BYTES = CH * 64, L_BYTES = LR * 64, STAGE = W_BYTES + L_BYTES, NSTG = 3;
constexpr int OFF_RING = RES_BYTES;
constexpr int OFF_TS = OFF_RING + NSTG * STAGE;     // 149504
constexpr int OFF_BIAS = OFF_TS + BM * 64;          // 157696
constexpr int OFF_GAMMA = OFF_BIAS + MAXS * CH * 2;
constexpr int OFF_BETA = OFF_GAMMA + CH * 2;
constexpr int OFF_SROW = OFF_BETA + CH * 2;
constexpr int LDS_TOTAL = OFF_SROW + 64;
static_assert(LDS_TOTAL <= 160 * 1024, "LDS budget");

struct Stage {
  const bf16_t *W, *bias, *Ad, *Bup;   // W [320][ldw], bias [320] or null, Ad [32][320] or null (no LoRA), Bup [320][32]
  bf16_t *T, *Ts;                      // [M][32], rows >= row0
  const bf16_t* res;                   // KEEP: residual rows [M][ldr] or null
  bf16_t* out;                         // [M][ldo] or null
  bf16_t* nout;                        // KEEP + ln: LayerNorm output [M][ldn] (rows >= nout_row0) or null
  float* stats;                        // KEEP + ln: (mean, rstd) per row [M][2]
  const bf16_t *gamma, *beta;
  long ldw, ldr, ldo, ldn;
  float eps;
  int keep, ln, nout_row0;
};

struct Args {
  const bf16_t* X;   // [M][ldx] chain input
  long ldx;
  const bf16_t* S;   // [nsamples][32] scale rows
  int M, rps, row0, nstage;
  long long* trace;   // tools/trace_chain.py (AQL_CHAIN_TRACE_BUF): 32 cycle stamps per (block, wave 0 / wave 4); null in production
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
// at most 3 + n operations outstanding, n in {0, 10, 16, 36}: the stores of the previous stage's epilogue / row pass
__device__ __forceinline__ void wait_tiles(int n) {
  if (n == 0) wait_vm<3>();
  else if (n == 10) wait_vm<13>();
  else if (n == 16) wait_vm<19>();
  else wait_vm<39>();
}

__global__ __launch_bounds__(NTH, 2) void chain_kernel(const Args a) {
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
#define CH_STAMP() do { if (trc != nullptr && mark_ < 32) trc[mark_] = __builtin_readcyclecounter(); ++mark_; } while (0)
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
    *reinterpret_cast<uint4*>(lds + OFF_BIAS + g * CH * 2 + c * 16) = v;
  }
  for (int g = 0; g < a.nstage; ++g)
    if (a.st[g].ln && tid < CH / 8) {
      *reinterpret_cast<uint4*>(lds + OFF_GAMMA + tid * 16) = *reinterpret_cast<const uint4*>(a.st[g].gamma + tid * 8);
      *reinterpret_cast<uint4*>(lds + OFF_BETA + tid * 16) = *reinterpret_cast<const uint4*>(a.st[g].beta + tid * 8);
    }
  if (tid < 4)
    *reinterpret_cast<uint4*>(lds + OFF_SROW + tid * 16) =
        (lora_tile && a.S) ? *reinterpret_cast<const uint4*>(a.S + (long)(m0 / a.rps) * LR + tid * 8) : make_uint4(0u, 0u, 0u, 0u);

  // ---- the chain input tile -> resident region: 80 instructions of 1 KB, 10 per wavefront (K tile kt, 16-row block rb)
  {
    const __amdgpu_buffer_rsrc_t rsX = make_rsrc(a.X);
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      const int q = wave + 8 * i, kt = q >> 3, rb = q & 7;
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
    char* dst = lds + OFF_RING + wr * STAGE;
    const uint32_t soff = (uint32_t)kt * 64u, v0 = dr0 * ldb + dchunk;
    dma16(rsW, dst + wave * 1024, v0, soff);
    dma16(rsW, dst + (wave + 8) * 1024, v0 + 128u * ldb, soff);
    if (wave < 4) dma16(rsW, dst + (wave + 16) * 1024, v0 + 256u * ldb, soff);
    else dma16(rsL, dst + W_BYTES + (wave & 1) * 1024, (uint32_t)((wave & 1) * 16 + drow) * (uint32_t)(CH * 2) + dchunk, soff);   // a null descriptor zero-fills
    wr = ring_next(wr);
  };
  auto issue_up = [&](const __amdgpu_buffer_rsrc_t& rsB) __attribute__((always_inline)) {   // the Bup tile [320][32]
    char* dst = lds + OFF_RING + wr * STAGE;
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

  const int aoff = (wm * 32) * 64 + lo;            // A fragments: rows wm*32 + 16 i + l15 of a K tile
  const int boff = (wn * 160) * 64 + lo;           // W fragments: rows wn*160 + 16 j + l15
  const int loff = W_BYTES + (wn * 16) * 64 + lo;  // LoRA-down fragment: rank rows wn*16 + l15
  int pend = 0;                                    // row-pass stores of the previous stage still counted by vmcnt (per wavefront)

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

    f32x4_t acc[2][10], tacc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
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
        // iterations 0 / 1: the previous stage's row-pass stores are YOUNGER than this tile's requests -- count past them; from
        // iteration 2 on the awaited tile is younger than the stores, which have had two tiles' time to drain
        if (t < 2) wait_tiles(pend);
        else wait_vm<3>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t == 0) CH_STAMP();   // 1 + 6 g: the stage's first K tile (g = 0: and the chain input) has landed
        const char* sW = lds + OFF_RING + rd * STAGE;
        const char* sA = lds + t * (BM * 64);
        bf16x8_t fa[2], fb[10], fl;
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const bf16x8_t*>(sA + aoff + i * 1024);
#pragma unroll
        for (int j = 0; j < 10; ++j) fb[j] = *reinterpret_cast<const bf16x8_t*>(sW + boff + j * 1024);
        if constexpr (LORA) fl = *reinterpret_cast<const bf16x8_t*>(sW + loff);
#pragma unroll
        for (int j = 0; j < 10; ++j) acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[0], acc[0][j], 0, 0, 0);
        if constexpr (LORA) tacc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fl, fa[0], tacc[0], 0, 0, 0);
        issue_ahead(t + 2);
#pragma unroll
        for (int j = 0; j < 10; ++j) acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[1], acc[1][j], 0, 0, 0);
        if constexpr (LORA) tacc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fl, fa[1], tacc[1], 0, 0, 0);
        rd = ring_next(rd);
      }
    };
    if (lora_g) mainloop(std::true_type{});
    else mainloop(std::false_type{});
    pend = 0;
    CH_STAMP();   // 2 + 6 g: K loop issued

    if (lora_g) {
      // ---- T -> (T, Ts) bf16; Ts as one more A tile; one k-step against the Bup tile of the ring
      const uint2 sv = *reinterpret_cast<const uint2*>(lds + OFF_SROW + (wn * 16 + q4 * 4) * 2);
      const __amdgpu_buffer_rsrc_t rsT = make_rsrc(s.T), rsTs = make_rsrc(s.Ts);
      const uint32_t vt = (uint32_t)(m0 + wm * 32 + l15) * 64u + (uint32_t)(wn * 16 + q4 * 4) * 2u;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = wm * 32 + i * 16 + l15;
        const int r = wn * 16 + q4 * 4;
        const u32x2_t tv = {pack_bf16x2(tacc[i][0], tacc[i][1]), pack_bf16x2(tacc[i][2], tacc[i][3])};
        const u32x2_t ts = {pack_bf16x2(bf16lo(tv.x) * bf16lo(sv.x), bf16hi(tv.x) * bf16hi(sv.x)),
                            pack_bf16x2(bf16lo(tv.y) * bf16lo(sv.y), bf16hi(tv.y) * bf16hi(sv.y))};
        *reinterpret_cast<u32x2_t*>(lds + OFF_TS + off64(row, r >> 3) + (r & 7) * 2) = ts;
        __builtin_amdgcn_raw_buffer_store_b64(tv, rsT, vt, i * 1024, 0);
        __builtin_amdgcn_raw_buffer_store_b64(ts, rsTs, vt, i * 1024, 0);
      }
      wait_vm<7>();   // in order: [Bup tile 3] [next tile 3] [T / Ts stores 4]
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const char* sW = lds + OFF_RING + rd * STAGE;
      bf16x8_t fa[2], fb[10];
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const bf16x8_t*>(lds + OFF_TS + aoff + i * 1024);
#pragma unroll
      for (int j = 0; j < 10; ++j) fb[j] = *reinterpret_cast<const bf16x8_t*>(sW + boff + j * 1024);
#pragma unroll
      for (int j = 0; j < 10; ++j) acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[0], acc[0][j], 0, 0, 0);
      issue_ahead(NKT + 2);
#pragma unroll
      for (int j = 0; j < 10; ++j) acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[1], acc[1][j], 0, 0, 0);
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
      const char* sBias = lds + OFF_BIAS + g * CH * 2 + (wn * 160 + q4 * 4) * 2;
      const int c0 = wn * 160 + (q4 & 1) * 16 + (q4 >> 1) * 8;       // + 32 jp
      const __amdgpu_buffer_rsrc_t rsO = make_rsrc(keep ? nullptr : s.out);
      const uint32_t ldo2 = keep ? 0u : (uint32_t)(s.ldo * 2);
      const uint32_t vo = (uint32_t)(m0 + wm * 32 + l15) * ldo2 + (uint32_t)c0 * 2u;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = wm * 32 + i * 16 + l15;
        char* const ldst = lds + row * 64;
        const int sz = swz4(row);
#pragma unroll
        for (int jp = 0; jp < 5; ++jp) {
          const uint2 bA = *reinterpret_cast<const uint2*>(sBias + jp * 64), bB = *reinterpret_cast<const uint2*>(sBias + jp * 64 + 32);
          const f32x4_t& xa = acc[i][2 * jp];
          const f32x4_t& xb = acc[i][2 * jp + 1];
          uint32_t x0 = pack_bf16x2(xa[0] + bf16lo(bA.x), xa[1] + bf16hi(bA.x)), x1 = pack_bf16x2(xa[2] + bf16lo(bA.y), xa[3] + bf16hi(bA.y));
          uint32_t y0 = pack_bf16x2(xb[0] + bf16lo(bB.x), xb[1] + bf16hi(bB.x)), y1 = pack_bf16x2(xb[2] + bf16lo(bB.y), xb[3] + bf16hi(bB.y));
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
      pend = 10;   // 2 x 5 output stores per wavefront
      continue;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    CH_STAMP();   // 5 + 6 g: tile published

    // ---- row pass: wavefront w owns rows 16 w .. 16 w + 15, lanes 0..39 one 16-byte chunk each (whole 640-byte rows to HBM).
    // Branch-free: lanes 40..63 compute on chunk 0 and are switched off by out-of-range buffer offsets / a dummy LDS address.
    {
      const bool act = lane < CH / 8;
      const int cl = act ? lane : 0;
      const bool has_res = s.res != nullptr, has_out = s.out != nullptr, do_ln = s.ln != 0;
      const bool wr_n = do_ln && s.nout != nullptr && m0 >= s.nout_row0;    // block-uniform
      const float eps = s.eps;
      const uint32_t ldr2 = keep_s((uint32_t)(s.ldr * 2)), ldo2 = keep_s((uint32_t)(s.ldo * 2)), ldn2 = keep_s((uint32_t)(s.ldn * 2));
      const __amdgpu_buffer_rsrc_t rsR = make_rsrc(s.res), rsO = make_rsrc(s.out), rsN = make_rsrc(wr_n ? s.nout : nullptr),
                                   rsS = make_rsrc(do_ln ? s.stats : nullptr);
      const uint32_t mrow = (uint32_t)(m0 + wave * 16);
      const uint32_t vr = mrow * ldr2 + cl * 16;
      const uint32_t vo = act ? mrow * ldo2 + cl * 16 : OOB_ROW;
      const uint32_t vn = act ? mrow * ldn2 + cl * 16 : OOB_ROW;
      const uint32_t vs = lane == 0 ? mrow * 8u : OOB_ROW;
      const int lbase = (cl >> 2) * (BM * 64) + (wave * 16) * 64;
      uint4 gr = make_uint4(0u, 0u, 0u, 0u), br = gr;
      if (do_ln) {
        gr = *reinterpret_cast<const uint4*>(lds + OFF_GAMMA + cl * 16);
        br = *reinterpret_cast<const uint4*>(lds + OFF_BETA + cl * 16);
      }
      // lane (16 k + rr' ...) of a wave_sum8 result: the (mean, rstd) pair of row rr is stored by lane 16 * (0, 2, 1, 3)[rr & 3]
      const int myrow = ((q4 & 1) << 1) | (q4 >> 1);             // the row (of four) whose statistics this lane's 16-lane row holds
      const uint32_t vs4 = l15 == 0 ? (mrow + myrow) * 8u : OOB_ROW;
      auto batch = [&](auto res_tag, auto ln_tag, int r0) __attribute__((always_inline)) {
        constexpr bool RES = decltype(res_tag)::value, LN = decltype(ln_tag)::value;
        uint4 xv[8];
        u32x4_t rv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int rr = r0 + u;   // row wave*16 + rr: swizzle term g[(rr >> 2) & 3]
          if constexpr (RES) rv[u] = __builtin_amdgcn_raw_buffer_load_b128(rsR, vr, rr * ldr2, 0);
          xv[u] = *reinterpret_cast<const uint4*>(lds + lbase + rr * 64 + ((((cl & 3)) ^ swz4(rr)) << 4));
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int rr = r0 + u;
          if constexpr (RES) xv[u] = epi_add8(xv[u], make_uint4(rv[u].x, rv[u].y, rv[u].z, rv[u].w));
          __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{xv[u].x, xv[u].y, xv[u].z, xv[u].w}, rsO, vo + rr * ldo2, 0, 0);   // null descriptor: dropped
        }
        if constexpr (LN) {
          float ma, mb, ra, rb;
          ln_rows8(xv, act, gr, br, eps, ma, mb, ra, rb);
          __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{__float_as_uint(ma), __float_as_uint(ra)}, rsS, vs4, r0 * 8, 0);
          __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{__float_as_uint(mb), __float_as_uint(rb)}, rsS, vs4, (r0 + 4) * 8, 0);
#pragma unroll
          for (int u = 0; u < 8; ++u)
            __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{xv[u].x, xv[u].y, xv[u].z, xv[u].w}, rsN, vn + (r0 + u) * ldn2, 0, 0);
        }
        if constexpr (RES || LN) {
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int rr = r0 + u;
            char* const wp = act ? lds + lbase + rr * 64 + ((((cl & 3)) ^ swz4(rr)) << 4) : lds + OFF_TS + lane * 16;
            *reinterpret_cast<uint4*>(wp) = xv[u];
          }
        }
      };
      auto pass = [&](auto res_tag, auto ln_tag) __attribute__((always_inline)) {
        batch(res_tag, ln_tag, 0);
        batch(res_tag, ln_tag, 8);
      };
      if (has_res && do_ln) pass(std::true_type{}, std::true_type{});
      else if (do_ln) pass(std::false_type{}, std::true_type{});
      else if (has_res) pass(std::true_type{}, std::false_type{});
      else pass(std::false_type{}, std::false_type{});
      (void)has_out;
      (void)vs;
      pend = do_ln ? 36 : 16;   // stores ISSUED per wavefront (dropped ones included): 16 rows out, + 4 statistics + 16 rows nout
    }
    CH_STAMP();   // 6 + 6 g: row pass done
    // the next stage's first barrier publishes the rewritten tile
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // trailing zero-fill DMAs still target the LDS
  CH_STAMP();   // everything drained
#undef CH_STAMP
}

}  // namespace aqlchain

extern "C" int aql_lora_chain_fwd(const bf16_t* X, long ldx, long M, int rows_per_sample, long lora_row0, const bf16_t* S, int nstage,
                                  const void* const* W, const long* ldw, const void* const* bias, const void* const* Adown,
                                  const void* const* Bup, void* const* T, void* const* Ts, const void* const* res, const long* ldr,
                                  void* const* out, const long* ldo, const int* keep, const int* ln, const void* const* gamma,
                                  const void* const* beta, const float* eps, void* const* stats, void* const* nout, const long* ldn,
                                  const long* nout_row0, hipStream_t stream) {
  using namespace aqlchain;
  AQL_CHECK_ARG(X != nullptr && nstage >= 1 && nstage <= MAXS, "aql_lora_chain_fwd: 1..%d stages", MAXS);
  AQL_CHECK_ARG(M > 0 && M % BM == 0 && M < (1L << 30), "aql_lora_chain_fwd: M = %ld must be a multiple of %d", M, BM);
  AQL_CHECK_ARG(rows_per_sample > 0 && rows_per_sample % BM == 0, "aql_lora_chain_fwd: rows_per_sample %% %d != 0", BM);
  AQL_CHECK_ARG(lora_row0 >= 0 && lora_row0 % BM == 0, "aql_lora_chain_fwd: lora_row0 %% %d != 0", BM);
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
  {
    const char* tb = getenv("AQL_CHAIN_TRACE_BUF");   // device address of the stamp buffer (tools/trace_chain.py)
    a.trace = tb ? (long long*)strtoull(tb, nullptr, 0) : nullptr;
  }
  int nln = 0;
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
    s.ln = ln ? ln[g] : 0;
    s.gamma = gamma ? (const bf16_t*)gamma[g] : nullptr;
    s.beta = beta ? (const bf16_t*)beta[g] : nullptr;
    s.eps = eps ? eps[g] : 0.f;
    s.stats = stats ? (float*)stats[g] : nullptr;
    s.nout = nout ? (bf16_t*)nout[g] : nullptr;
    s.ldn = ldn ? ldn[g] : 0;
    s.nout_row0 = nout_row0 ? (int)nout_row0[g] : 0;
    AQL_CHECK_ARG(s.W != nullptr && s.ldw >= CH, "aql_lora_chain_fwd: stage %d has no weight", g);
    AQL_CHECK_ARG(s.Ad == nullptr || (s.Bup && s.T && s.Ts && S), "aql_lora_chain_fwd: stage %d: LoRA needs Bup, T, Ts and S", g);
    AQL_CHECK_ARG(s.keep || (s.out != nullptr && s.res == nullptr && !s.ln), "aql_lora_chain_fwd: stage %d: a DIRECT stage writes `out` only", g);
    AQL_CHECK_ARG(!s.ln || (s.keep && s.gamma && s.beta && s.stats), "aql_lora_chain_fwd: stage %d: LayerNorm needs keep, gamma, beta, stats", g);
    AQL_CHECK_ARG(s.out == nullptr || (s.ldo >= CH && M * s.ldo * 2 < (long)BUF_BYTES), "aql_lora_chain_fwd: stage %d: output span", g);
    AQL_CHECK_ARG(s.res == nullptr || (s.ldr >= CH && M * s.ldr * 2 < (long)BUF_BYTES), "aql_lora_chain_fwd: stage %d: residual span", g);
    AQL_CHECK_ARG(s.nout == nullptr || (s.ldn >= CH && M * s.ldn * 2 < (long)BUF_BYTES && s.nout_row0 % BM == 0), "aql_lora_chain_fwd: stage %d: LayerNorm output span / first row", g);
    nln += s.ln ? 1 : 0;
  }
  AQL_CHECK_ARG(nln <= 1, "aql_lora_chain_fwd: at most one LayerNorm per chain");
  static bool once = false;
  if (!once) {
    (void)hipFuncSetAttribute((const void*)chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
    once = true;
  }
  hipLaunchKernelGGL(chain_kernel, dim3((unsigned)(M / BM)), dim3(NTH), LDS_TOTAL, stream, a);
  AQL_CHECK_LAUNCH("aql_lora_chain_fwd");
  return AQL_OK;
}
