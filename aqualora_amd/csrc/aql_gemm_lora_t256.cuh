// ff.net.0 + rank-32 LoRA + GEGLU on a 256 x 256 workgroup tile (round 4): the one-launch LoRA linear of aql_gemm_lora.hip
// (utils/lora_modules.py:9-26,56-62 + scripts/lib/original_unet.py:727-729) re-tiled for the shapes where the 128 x 160 kernel is
// paced by its L2 -> LDS traffic and by the store burst of its GEGLU epilogue (32768 x 2560 x 320: 117 us, roofline.frac 0.19).
//
//   * 256 rows x [128 value | 128 gate] columns per tile, EIGHT wavefronts (2 along M x 4 along N), each 128 rows x (32 value + 32
//     gate columns of the SAME 32 features): the GEGLU is lane-local, no tile exchange between the halves.
//   * K loop in four phases per 64-wide K tile (one 64 x 32 quadrant each), the LDS-DMA of one half tile of the next K tile at the top
//     of every phase; two 68 KB stages (X 256 x 64, W 256 x 64, LoRA-down 32 x 64).  Half the bytes per FLOP of the 128 x 160 tile
//     through the ~60 B/clk/CU DMA path.  (tools/micro/gemm256.hip is the stand-alone prototype of this loop: 1235 TFLOP/s at 8192^3.)
//   * PERSISTENT, one workgroup per CU, and the wavefronts split their memory roles: 0-3 issue every load, 4-7 every global store.
//     vmcnt retires in order per wavefront (stores included on gfx950): a wavefront that stored tile i would wait for that drain at
//     its first DMA wait of tile i + 1; split like this the loaders never have a store outstanding and the storers never wait.
//   * outputs leave through LDS as whole 256-byte row segments (G | H-value in one pass, H-gate in a second); bias and the Bup panel
//     arrive by LDS-DMA in a spare 17 KB, so nothing but the scale rows is held in registers across the K loop.
//   * the rank-32 side product T = X A^T rides in the K loop (each wavefront owns 64 rows x 16 rank columns: 8 MFMAs per K tile),
//     T / Ts are written by the first column tile, the up-projection is one extra k-step -- same operations in the same order as
//     lora_gemm_kernel: BIT-IDENTICAL outputs (tools/probe_lora_persist.py PROBE_ALT=t256).
#pragma once
#include <type_traits>
#include "aql_gemm.cuh"

namespace aqlt256 {
using namespace aqlgemm;

constexpr int TM = 256, NTH = 512, LR_ = 32;
constexpr int ST_B = 32768, ST_L = 65536, STAGE = 65536 + 4096;
constexpr int OFF_BUP = 2 * STAGE;          // Bup panel image: 256 rows x 64 B, 16-B chunk index XOR (row >> 2) & 3
constexpr int OFF_BIAS = OFF_BUP + 16384;   // bias image: 256 bf16 (value | gate)
constexpr int OFF_S = OFF_BIAS + 512;        // scale rows of the (at most 16) samples a tile's rows belong to, 64 B each
constexpr int LDS_TOTAL = OFF_S + 1024;
static_assert(LDS_TOTAL <= 160 * 1024, "LDS budget");

struct Args {
  const bf16_t *X, *W, *Ad, *S, *Bup, *bias;
  bf16_t *H, *G, *T, *Ts;
  long ldx, ldw, ldh, ldg;
  int M, K, F;
  int rps, row0, c_row0, ntiles;
  // SEG2 kernels (any LoRA rank; aql_gemm_bf16_geglu): a second K segment  + X2[M, K2] . W2[2F, K2]^T  on rows >= row0 (X2 = the scaled
  // LoRA down product Ts, W2 = Bup), no side product, no up step
  const bf16_t *X2, *W2;
  long ldx2, ldw2;
  int K2, seg2;
  long long* trace;   // -DAQL_T256_TRACE builds only (tools/trace_t256.py)
};

// output staging: row r, 16-byte chunk c of a 256-byte row segment at r * 256 + ((c ^ (r & 15)) << 4)
__device__ __forceinline__ int seg_off(int row, int c) { return row * 256 + ((c ^ (row & 15)) << 4); }
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

template <bool SEG2>
__global__ __launch_bounds__(NTH, 2) void lora_geglu256_kernel(const Args a) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  int tid_ = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid_ >> 6);
  const int wm = wave >> 2, wn = wave & 3, lw = wave & 3;
  const bool loader = wave < 4;
  const int tiles_n = a.F / 128, tiles_m = (a.M + TM - 1) / TM;
  const bool twin_mix = a.row0 > 0 && (tiles_m & 1) == 0 && a.row0 == (tiles_m >> 1) * TM;
  const int KT1 = (a.K + BK - 1) / BK, KT2 = SEG2 ? (a.K2 + BK - 1) / BK : 0;
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.X), 0, (uint32_t)a.M * (uint32_t)(a.ldx * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.W), 0, (uint32_t)(2 * a.F) * (uint32_t)(a.ldw * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsL = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.Ad), 0, (uint32_t)LR_ * (uint32_t)(a.K * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsU = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.Bup), 0, (uint32_t)(2 * a.F) * 64u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.S), 0, (uint32_t)((a.M + a.rps - 1) / a.rps) * 64u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsBi = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.bias), 0, a.bias ? (uint32_t)(2 * a.F) * 2u : 0u, 0x00020000);
  // outputs through buffer stores: rows past M fall outside the descriptor, masked rows get an out-of-range offset -- the store
  // INSTRUCTION is always issued, so the number of stores a wavefront has in flight is a compile-time constant (counted vmcnt below)
  const __amdgpu_buffer_rsrc_t rsG = __builtin_amdgcn_make_buffer_rsrc(a.G, 0, (uint32_t)a.M * (uint32_t)(a.ldg * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsH = __builtin_amdgcn_make_buffer_rsrc(a.H, 0, a.H ? (uint32_t)a.M * (uint32_t)(a.ldh * 2) : 0u, 0x00020000);
  const uint32_t stepX = 32u * (uint32_t)(a.ldx * 2), stepW = 32u * (uint32_t)(a.ldw * 2);
  const __amdgpu_buffer_rsrc_t rsX2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.X2), 0, SEG2 ? (uint32_t)a.M * (uint32_t)(a.ldx2 * 2) : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.W2), 0, SEG2 ? (uint32_t)(2 * a.F) * (uint32_t)(a.ldw2 * 2) : 0u, 0x00020000);
  const uint32_t stepX2 = SEG2 ? 32u * (uint32_t)(a.ldx2 * 2) : 0u, stepW2 = SEG2 ? 32u * (uint32_t)(a.ldw2 * 2) : 0u;
  const int aBase = (wm * 128) * 128;
  const int bBaseV = ST_B + (wn * 32) * 128, bBaseG = ST_B + (128 + wn * 32) * 128;   // this wavefront's value / gate weight rows
  const int lBase = ST_L + ((wn & 1) * 16) * 128;                                     // its 16 rank rows of the LoRA-down tile
  const int tp = wn >> 1;                                                             // which 64-row half its T rows belong to

  // tile coordinates (all scalar)
  struct Tile { int m0, f0, kt; bool lora_on, t_writer, want_h; uint32_t rowX, rowWv, rowWg; };
  auto tile_of = [&](int tl) __attribute__((always_inline)) {
    Tile c;
    int L = tl;
    if ((a.ntiles & 7) == 0) L = (tl & 7) * (a.ntiles >> 3) + (tl >> 3);   // tile tl runs on XCD tl % 8: contiguous logical runs per XCD
    int tile_m = L / tiles_n;
    const int tile_n = L - tile_m * tiles_n;
    // twin batch (rows below row0 = the clean half: no LoRA side product, no H): a tile of the second half costs ~1.4x a tile of the
    // first, and a contiguous run of logical tiles per XCD would give XCDs 0-3 only cheap tiles.  Alternate the halves row tile by row tile.
    if (twin_mix) tile_m = (tile_m & 1) ? (tiles_m >> 1) + (tile_m >> 1) : (tile_m >> 1);
    c.m0 = tile_m * TM, c.f0 = tile_n * 128;
    c.lora_on = !SEG2 && c.m0 + TM > a.row0;
    c.kt = KT1 + (SEG2 && c.m0 + TM > a.row0 ? KT2 : 0);     // tiles of the clean half skip the second segment
    c.t_writer = tile_n == 0;
    c.want_h = a.H != nullptr && c.m0 + TM > a.c_row0;
    c.rowX = (uint32_t)c.m0 * (uint32_t)(a.ldx * 2);
    c.rowWv = (uint32_t)c.f0 * (uint32_t)(a.ldw * 2), c.rowWg = (uint32_t)(a.F + c.f0) * (uint32_t)(a.ldw * 2);
    return c;
  };
  // The K loop of every tile starts in REGION 1 (lds + STAGE): the output staging of the previous tile's last pass lives in
  // [0, 64 KB), so the loaders can request the next tile's first K tile, Bup panel, bias and scale rows while that pass drains.
  auto region = [&](int t) __attribute__((always_inline)) { return lds + ((t + 1) & 1) * STAGE; };

  // loader lanes (rebuilt from a laundered thread id wherever they are used: hoisted out of the persistent loop the lane-derived
  // addresses cost ~20 VGPRs that nothing in the K loop can spare)
  struct LoadLane { int kc, lrow; uint32_t vX, vW, vL, vX2, vW2; };
  auto load_lane = [&](int tid) __attribute__((always_inline)) {
    LoadLane q;
    const int lane = tid & 63;
    // a half tile (128 rows x 64) is 16 instructions of 8 rows; loader w issues instructions w, w + 4, w + 8, w + 12.  Row inside the
    // half = (4 i + w) * 8 + (lane >> 3): its swizzle term (row >> 1) & 7 does not depend on i.
    const int lrow = lw * 8 + (lane >> 3);
    const int lchunk = (lane & 7) ^ ((lrow >> 1) & 7);
    q.kc = lchunk * 8;
    q.vX = (uint32_t)lrow * (uint32_t)(a.ldx * 2) + lchunk * 16;
    q.vW = (uint32_t)lrow * (uint32_t)(a.ldw * 2) + lchunk * 16;
    q.vL = (uint32_t)lrow * (uint32_t)(a.K * 2) + lchunk * 16;          // LoRA-down rows 8 w .. 8 w + 7
    q.lrow = lrow;
    q.vX2 = SEG2 ? (uint32_t)lrow * (uint32_t)(a.ldx2 * 2) + lchunk * 16 : 0u;
    q.vW2 = SEG2 ? (uint32_t)lrow * (uint32_t)(a.ldw2 * 2) + lchunk * 16 : 0u;
    return q;
  };
  auto dma_x = [&](const LoadLane& q, const Tile& c, int half, int t, char* stage) __attribute__((always_inline)) {
    if (SEG2 && t >= KT1) {      // second segment: rows below row0 (the clean half of a twin batch) contribute nothing
      const int t2 = t - KT1;
      const bool bad = q.kc >= a.K2 - t2 * BK;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool below = c.m0 + half * 128 + i * 32 + q.lrow < a.row0;
        dma16(rsX2, stage + half * 16384 + (i * 4 + lw) * 1024,
              (bad | below) ? OOB_ROW : q.vX2 + (uint32_t)c.m0 * (uint32_t)(a.ldx2 * 2) + (half * 4 + i) * stepX2, (uint32_t)t2 * (BK * 2));
      }
      return;
    }
    const bool bad = q.kc >= a.K - t * BK;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      dma16(rsX, stage + half * 16384 + (i * 4 + lw) * 1024, bad ? OOB_ROW : q.vX + c.rowX + (half * 4 + i) * stepX, (uint32_t)t * (BK * 2));
  };
  auto dma_w = [&](const LoadLane& q, const Tile& c, int half, int t, char* stage) __attribute__((always_inline)) {
    if (SEG2 && t >= KT1) {
      const int t2 = t - KT1;
      const bool bad = q.kc >= a.K2 - t2 * BK;
      const uint32_t roww = (uint32_t)(half ? a.F + c.f0 : c.f0) * (uint32_t)(a.ldw2 * 2);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        dma16(rsW2, stage + ST_B + half * 16384 + (i * 4 + lw) * 1024, bad ? OOB_ROW : q.vW2 + roww + i * stepW2, (uint32_t)t2 * (BK * 2));
      return;
    }
    const bool bad = q.kc >= a.K - t * BK;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      dma16(rsW, stage + ST_B + half * 16384 + (i * 4 + lw) * 1024, bad ? OOB_ROW : q.vW + (half ? c.rowWg : c.rowWv) + i * stepW, (uint32_t)t * (BK * 2));
  };
  auto dma_l = [&](const LoadLane& q, const Tile& c, int t, char* stage) __attribute__((always_inline)) {
    if constexpr (SEG2) return;
    const bool bad = (q.kc >= a.K - t * BK) | !c.lora_on;
    dma16(rsL, stage + ST_L + lw * 1024, bad ? OOB_ROW : q.vL, (uint32_t)t * (BK * 2));
  };
  // tile prologue (loaders): K tile 0, the Bup panel, the bias, the scale rows of the tile's samples
  auto prologue = [&](const Tile& c, int tid) __attribute__((always_inline)) {
    const LoadLane q = load_lane(tid);
    const int lane = tid & 63;
    char* st = region(0);
    dma_x(q, c, 0, 0, st);
    dma_x(q, c, 1, 0, st);
    dma_w(q, c, 0, 0, st);
    dma_w(q, c, 1, 0, st);
    dma_l(q, c, 0, st);
    if (c.lora_on) {
      // image row b <-> output column: b < 128 value feature f0 + b, else gate feature f0 + b - 128; 16 rows (64 B each) per instruction
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int b = (i * 4 + lw) * 16 + (lane >> 2);
        const int n = b < 128 ? c.f0 + b : a.F + c.f0 + b - 128;
        const int ch = (lane & 3) ^ ((b >> 2) & 3);
        dma16(rsU, lds + OFF_BUP + (i * 4 + lw) * 1024, (uint32_t)n * 64u + ch * 16, 0);
      }
      if (lw == 1) dma16(rsS, lds + OFF_S, (uint32_t)(c.m0 / a.rps) * 64u + lane * 16, 0);   // samples m0 / rps .. + 15
    }
    if (lw == 0) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsBi, (__attribute__((address_space(3))) void*)(lds + OFF_BIAS + h * 256), 4,
                                                 (uint32_t)((h ? a.F + c.f0 : c.f0) * 2 + lane * 4), 0, 0, 0);
    }
  };

#ifdef AQL_T256_TRACE
  int mark_ = 0;
#define T256_STAMP() do { if (a.trace != nullptr && (blockIdx.x == 0 || blockIdx.x == 77) && (tid_ & 63) == 0 && (wave == 0 || wave == 4) && mark_ < 96) \
    a.trace[((blockIdx.x != 0) * 2 + (wave >> 2)) * 96 + mark_] = __builtin_readcyclecounter(); ++mark_; } while (0)
#else
#define T256_STAMP() do { } while (0)
#endif
  if (blockIdx.x < a.ntiles && loader) prologue(tile_of(blockIdx.x), tid_);
  for (int tl = blockIdx.x; tl < a.ntiles; tl += gridDim.x) {
    asm volatile("" : "+v"(tid_));
    const int tid = tid_, lane = tid & 63;
    T256_STAMP();   // 0 tile start
    const Tile c = tile_of(tl);
    const int m0 = c.m0, f0 = c.f0;
    const bool lora_on = c.lora_on, t_writer = c.t_writer, want_h = c.want_h;
    const LoadLane q = load_lane(tid);
    // fragment reads: row = base + 16 f + (lane & 15), so (row >> 1) & 7 = (lane & 15) >> 1 for every fragment
    const int swz = (lane & 15) >> 1;
    const int offk0 = (lane & 15) * 128 + (((lane >> 4) ^ swz) << 4);
    const int offk1 = (lane & 15) * 128 + (((4 + (lane >> 4)) ^ swz) << 4);

    f32x4_t acc[8][4], tacc[4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) tacc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // this tile's prologue was requested under the previous tile's epilogue, BEFORE that tile's 8 G stores: loads and stores retire
    // in issue order, so "at most 8 outstanding" means every prologue load has landed while the G stores may still be draining
    if (loader) {
      if (tl == (int)blockIdx.x) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef AQL_T256_TRACE
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the trace stamps are stores too
#else
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
#endif
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    T256_STAMP();   // 1 first K tile there

    // ---- K loop
    bf16x8_t fa[4][2], fb[2][2];   // one 64-row half of A, one 32-column half of W at a time (the value half is read twice per K tile)
    auto read_a = [&](const char* st, int mq) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fa[i][0] = *reinterpret_cast<const bf16x8_t*>(st + aBase + (mq * 4 + i) * 2048 + offk0);
        fa[i][1] = *reinterpret_cast<const bf16x8_t*>(st + aBase + (mq * 4 + i) * 2048 + offk1);
      }
    };
    auto read_b = [&](const char* st, int nq) __attribute__((always_inline)) {   // nq 0: the 32 value columns, 1: the 32 gate columns
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        fb[j][0] = *reinterpret_cast<const bf16x8_t*>(st + (nq ? bBaseG : bBaseV) + j * 2048 + offk0);
        fb[j][1] = *reinterpret_cast<const bf16x8_t*>(st + (nq ? bBaseG : bBaseV) + j * 2048 + offk1);
      }
    };
    auto quad = [&](int mq, int nq) __attribute__((always_inline)) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[mq * 4 + i][nq * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j][ks], fa[i][ks], acc[mq * 4 + i][nq * 2 + j], 0, 0, 0);
    };
    auto side = [&](const char* st) __attribute__((always_inline)) {            // T rows of the 64-row half whose fragments are in `fa` right now
      const bf16x8_t l0 = *reinterpret_cast<const bf16x8_t*>(st + lBase + offk0);
      const bf16x8_t l1 = *reinterpret_cast<const bf16x8_t*>(st + lBase + offk1);
#pragma unroll
      for (int i = 0; i < 4; ++i) tacc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(l0, fa[i][0], tacc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) tacc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(l1, fa[i][1], tacc[i], 0, 0, 0);
    };
    // TP: -1 no LoRA in this tile, 0 / 1 = the half this wavefront's T rows belong to (a straight-line loop per case: a branch around
    // the side MFMAs inside the K tile cuts the compiler's ds_read / MFMA interleave)
    auto kloop = [&](auto tp_tag) __attribute__((always_inline)) {
      constexpr int TP = decltype(tp_tag)::value;
      const int KT = c.kt;
      for (int t = 0; t < KT; ++t) {
        char* cur = region(t);
        char* nxt = region(t + 1);
        const bool more = loader && t + 1 < KT;
        if (more) dma_x(q, c, 0, t + 1, nxt);
        read_b(cur, 0);
        read_a(cur, 0);
        quad(0, 0);
        if constexpr (TP == 0) side(cur);
        if (more) dma_x(q, c, 1, t + 1, nxt);
        read_b(cur, 1);
        quad(0, 1);
        if (more) {
          dma_w(q, c, 0, t + 1, nxt);
          dma_l(q, c, t + 1, nxt);
        }
        read_a(cur, 1);
        quad(1, 1);
        if constexpr (TP == 1) side(cur);
        if (more) dma_w(q, c, 1, t + 1, nxt);
        read_b(cur, 0);
        quad(1, 0);
        if (loader) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
    };
    if constexpr (SEG2) {
      kloop(std::integral_constant<int, -1>{});
    } else {
      if (!lora_on) kloop(std::integral_constant<int, -1>{});
      else if (tp == 0) kloop(std::integral_constant<int, 0>{});
      else kloop(std::integral_constant<int, 1>{});
    }

    T256_STAMP();   // 2 K loop done
    // ---- LoRA: T -> (T, Ts) bf16, Ts as an A image at the bottom of the LDS, one k-step against the Bup panel
    if (!SEG2 && lora_on) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = wm * 128 + tp * 64 + i * 16 + (lane & 15);
        const long m = (long)m0 + row;
        const int r = (wn & 1) * 16 + (lane >> 4) * 4;
        const uint2 tv = make_uint2(pack_bf16x2(tacc[i][0], tacc[i][1]), pack_bf16x2(tacc[i][2], tacc[i][3]));
        const int si = (int)((uint32_t)(m < a.M ? m : a.M - 1) / (uint32_t)a.rps) - m0 / a.rps;
        const uint2 sv = *reinterpret_cast<const uint2*>(lds + OFF_S + si * 64 + r * 2);
        const uint2 ts = make_uint2(pack_bf16x2(bf16lo(tv.x) * bf16lo(sv.x), bf16hi(tv.x) * bf16hi(sv.x)),
                                    pack_bf16x2(bf16lo(tv.y) * bf16lo(sv.y), bf16hi(tv.y) * bf16hi(sv.y)));
        *reinterpret_cast<uint2*>(lds + lds_off(row, r >> 3) + (r & 7) * 2) = ts;
        if (m < a.M && t_writer) {
          *reinterpret_cast<uint2*>(a.T + m * LR_ + r) = tv;
          *reinterpret_cast<uint2*>(a.Ts + m * LR_ + r) = ts;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      bf16x8_t ft[8], fu[4];
#pragma unroll
      for (int i = 0; i < 8; ++i) ft[i] = *reinterpret_cast<const bf16x8_t*>(lds + aBase + i * 2048 + offk0);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int b = (j >> 1) * 128 + wn * 32 + (j & 1) * 16 + (lane & 15);
        fu[j] = *reinterpret_cast<const bf16x8_t*>(lds + OFF_BUP + b * 64 + (((lane >> 4) ^ ((b >> 2) & 3)) << 4));
      }
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fu[j], ft[i], acc[i][j], 0, 0, 0);
    }
    // ---- bias (fp32 add), round to bf16: h[i][0..1] / h[i][2..3] = this lane's 4 consecutive value / gate columns of 8 x 2 fragments
    uint2 bz[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      bz[j] = *reinterpret_cast<const uint2*>(lds + OFF_BIAS + ((j >> 1) * 128 + wn * 32 + (j & 1) * 16 + (lane >> 4) * 4) * 2);
    uint2 h[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        h[i][j] = make_uint2(pack_bf16x2(acc[i][j][0] + bf16lo(bz[j].x), acc[i][j][1] + bf16hi(bz[j].x)),
                             pack_bf16x2(acc[i][j][2] + bf16lo(bz[j].y), acc[i][j][3] + bf16hi(bz[j].y)));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();     // every wavefront has read its Ts / Bup / bias fragments: the LDS becomes the output staging
    asm volatile("" ::: "memory");

    T256_STAMP();   // 3 up step, bias, rounding done; LDS free
    const int tl_next = tl + gridDim.x;
    uint2 gq[8][2];
    auto geglu_regs = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          uint32_t g0 = geglu_word(h[i][jj].x, h[i][2 + jj].x), g1 = geglu_word(h[i][jj].y, h[i][2 + jj].y);
          asm volatile("" : "+v"(g0), "+v"(g1));   // computed HERE (under the storers' H stores), not sunk behind the barrier
          gq[i][jj] = make_uint2(g0, g1);
        }
    };
    auto stage_out = [&](char* base, auto&& pick) __attribute__((always_inline)) {   // this wavefront's 128 x 32 block of a 256-byte-row output segment image
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = wm * 128 + i * 16 + (lane & 15);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const int col = wn * 32 + jj * 16 + (lane >> 4) * 4;     // feature column inside the tile
          *reinterpret_cast<uint2*>(base + seg_off(row, col >> 3) + (col & 7) * 2) = pick(i, jj);
        }
      }
    };
    // all 8 wavefronts write whole 256-byte row segments: 8 buffer stores per wavefront and 64 KB image
    auto rows_out = [&](const char* base, const __amdgpu_buffer_rsrc_t& rs, long ld, int col0, int row_lo) __attribute__((always_inline)) {
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int id = it * 512 + tid, row = id >> 4, ch = id & 15;
        const int m = m0 + row;
        const u32x4_t v = *reinterpret_cast<const u32x4_t*>(base + seg_off(row, ch));
        const uint32_t off = m >= row_lo ? (uint32_t)m * (uint32_t)(ld * 2) + (uint32_t)(col0 + ch * 8) * 2u : OOB_ROW;
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 0);
      }
    };
#define AQL_T256_BAR() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
    if (want_h) {
      // ---- pass H: value half at 0, gate half at 64 KB; the loaders compute their GEGLU under the storers' 128 KB of stores
      stage_out(lds, [&](int i, int jj) __attribute__((always_inline)) { return h[i][jj]; });
      stage_out(lds + 65536, [&](int i, int jj) __attribute__((always_inline)) { return h[i][2 + jj]; });
      AQL_T256_BAR();
      T256_STAMP();   // 4 H staged
      rows_out(lds, rsH, a.ldh, f0, a.c_row0);
      rows_out(lds + 65536, rsH, a.ldh, a.F + f0, a.c_row0);
      T256_STAMP();   // 5 H stores issued (storers)
      geglu_regs();
      T256_STAMP();   // 6 GEGLU in registers
      AQL_T256_BAR();      // the storers' LDS reads are done (their global stores need not be)
    } else {
      T256_STAMP();
      T256_STAMP();
      geglu_regs();
      T256_STAMP();
    }
    T256_STAMP();     // 7 barrier
    // the next tile's prologue flies into region 1 / the spare images under pass G (which stages in [0, 64 KB) only)
    if (loader && tl_next < a.ntiles) prologue(tile_of(tl_next), tid);
    // ---- pass G
    stage_out(lds, [&](int i, int jj) __attribute__((always_inline)) { return gq[i][jj]; });
    AQL_T256_BAR();
    T256_STAMP();     // 8 G staged
    rows_out(lds, rsG, a.ldg, f0, 0);     // 8 stores per wavefront: the count the next tile's first wait leaves outstanding
    T256_STAMP();     // 9 G stores issued
    AQL_T256_BAR();
    T256_STAMP();     // 10 end
#undef AQL_T256_BAR
  }
}

inline void launch(const Args& a, hipStream_t stream) {
  static bool once = false;
  if (!once) {
    (void)hipFuncSetAttribute((const void*)lora_geglu256_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
    (void)hipFuncSetAttribute((const void*)lora_geglu256_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
    once = true;
  }
  const int grid = a.ntiles < 256 ? a.ntiles : 256;
  if (a.seg2) hipLaunchKernelGGL(lora_geglu256_kernel<true>, dim3(grid), dim3(NTH), LDS_TOTAL, stream, a);
  else hipLaunchKernelGGL(lora_geglu256_kernel<false>, dim3(grid), dim3(NTH), LDS_TOTAL, stream, a);
}

// aql_gemm_bf16_geglu's entry to the SEG2 kernel (defined in aql_gemm_lora.hip, the translation unit that instantiates the kernels):
// AQL_OK when the 256 x 256 tile took the launch, AQL_NOT_FUSED when the shape stays on the 128 x 160 kernels.
int t256_geglu_two_segments(const bf16_t* A, long lda, const bf16_t* B, long ldb, long M, int F, int K, const bf16_t* A2, long lda2,
                            const bf16_t* B2, long ldb2, int K2, const bf16_t* bias, bf16_t* H, long ldh, bf16_t* G, long ldg,
                            long row0, hipStream_t stream);

}  // namespace aqlt256
