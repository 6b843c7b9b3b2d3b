// GroupNorm(+SiLU) and LayerNorm, forward and backward-data, on channels-last bf16 activations.
// HBM-bound kernels: 16-byte vector loads, fp32 statistics, wavefront (64-lane) reductions.
// Reference semantics: torch.nn.GroupNorm(32, C, eps) / F.silu / torch.nn.LayerNorm(C) as used by the U-Net
// twin (scripts/lib/original_unet.py:423-453, 779-783, 826).  Base weights are frozen in PPFT, so backward
// produces the input gradient only.
#include "aql_common.h"

namespace {

constexpr int G = 32;  // NORM_GROUPS

__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
  f[0] = bf16lo(v.x); f[1] = bf16hi(v.x); f[2] = bf16lo(v.y); f[3] = bf16hi(v.y);
  f[4] = bf16lo(v.z); f[5] = bf16hi(v.z); f[6] = bf16lo(v.w); f[7] = bf16hi(v.w);
}
__device__ __forceinline__ uint4 pack8(const float* f) {
  return make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7]));
}
// v_rcp_f32 (1 ulp) instead of the IEEE division sequence (v_div_scale x2, v_rcp, 5 FMAs, v_div_fmas, v_div_fixup: ~10 of the
// ~27 VALU instructions per element of the GroupNorm + SiLU apply pass, which is VALU-bound, not HBM-bound: tools/tune_gn.py);
// the result is rounded to bf16 after one more multiply, three decimal digits above the difference
#ifdef AQL_SIGMOID_DIV   // A/B build (tools/build_alt.sh): the IEEE division
__device__ __forceinline__ float sigmoidf_(float z) { return 1.f / (1.f + __expf(-z)); }
#else
__device__ __forceinline__ float sigmoidf_(float z) { return __builtin_amdgcn_rcpf(1.f + __expf(-z)); }
#endif

// ---------------------------------------------------------------------------------------------------
// GroupNorm pass 1: per-(sample, group) partial sums over a slab of pixels.
// Block = cols*rp threads (cols = C/8 chunk columns, rp pixel rows in flight); thread owns one chunk column.
// MODE 0 (forward):  s1 = sum x,        s2 = sum x^2
// MODE 1 (backward): s1 = sum dxhat,    s2 = sum dxhat*xhat   with dxhat = dy*silu'(z)*gamma
// partial layout: [B][nslab][G][2]
// ---------------------------------------------------------------------------------------------------
template <int MODE>
__global__ void gn_partial_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                  const bf16_t* __restrict__ gamma, const bf16_t* __restrict__ beta,
                                  const float* __restrict__ stats, int HW, int C, int rows_per_slab, int silu,
                                  float* __restrict__ partial) {
  // per-thread partials are parked in LDS and reduced in a fixed order: deterministic (no float atomics)
  extern __shared__ __attribute__((aligned(16))) float sp[];  // [blockDim.x][16]
  const int cols = C >> 3;
  const int rp = blockDim.x / cols;
  const int col = threadIdx.x % cols;
  const int rr = threadIdx.x / cols;
  const int b = blockIdx.y, slab = blockIdx.x;
  const int cpg = C / G;
  float a1[8], a2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) a1[j] = a2[j] = 0.f;
  float ga[8], be[8], mu[8], rs[8];
  if (MODE == 1) {
    unpack8(*reinterpret_cast<const uint4*>(gamma + col * 8), ga);
    unpack8(*reinterpret_cast<const uint4*>(beta + col * 8), be);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int g = (col * 8 + j) / cpg;
      mu[j] = stats[(b * G + g) * 2 + 0];
      rs[j] = stats[(b * G + g) * 2 + 1];
    }
  }
  const int r0 = slab * rows_per_slab;
  const int r1 = min(HW, r0 + rows_per_slab);
  for (int r = r0 + rr; r < r1; r += rp) {
    const long off = ((long)b * HW + r) * C + col * 8;
    float xv[8];
    unpack8(*reinterpret_cast<const uint4*>(x + off), xv);
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        a1[j] += xv[j];
        a2[j] += xv[j] * xv[j];
      }
    } else {
      float dv[8];
      unpack8(*reinterpret_cast<const uint4*>(dy + off), dv);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xh = (xv[j] - mu[j]) * rs[j];
        float d = dv[j];
        if (silu) {
          const float z = xh * ga[j] + be[j];
          const float s = sigmoidf_(z);
          d *= s * (1.f + z * (1.f - s));
        }
        d *= ga[j];
        a1[j] += d;
        a2[j] += d * xh;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    sp[threadIdx.x * 16 + j] = a1[j];
    sp[threadIdx.x * 16 + 8 + j] = a2[j];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < G * 2; i += blockDim.x) {
    const int g = i >> 1, which = i & 1;
    float acc = 0.f;
    for (int c = g * cpg; c < (g + 1) * cpg; ++c)
      for (int q = 0; q < rp; ++q) acc += sp[(q * cols + (c >> 3)) * 16 + which * 8 + (c & 7)];
    partial[(((long)b * gridDim.x + slab) * G) * 2 + i] = acc;
  }
}

// pass 1.5: reduce slabs.  MODE 0 -> stats = (mean, rstd);  MODE 1 -> (mean dxhat, mean dxhat*xhat)
template <int MODE>
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ partial, int nslab, float inv_count,
                                                          float eps, float* __restrict__ out) {
  // 256 threads: (group,which) = t & 63, slab quarter = t >> 6; fixed-order tree -> deterministic
  __shared__ float red[4][64];
  const int b = blockIdx.x, gw = threadIdx.x & 63, part = threadIdx.x >> 6;
  float acc = 0.f;
  for (int s = part; s < nslab; s += 4) acc += partial[((long)b * nslab + s) * (G * 2) + gw];
  red[part][gw] = acc;
  __syncthreads();
  if (threadIdx.x >= G) return;
  const int g = threadIdx.x;
  const float s1 = (red[0][2 * g] + red[1][2 * g]) + (red[2][2 * g] + red[3][2 * g]);
  const float s2 = (red[0][2 * g + 1] + red[1][2 * g + 1]) + (red[2][2 * g + 1] + red[3][2 * g + 1]);
  if (MODE == 0) {
    const float mean = s1 * inv_count;
    const float var = fmaxf(s2 * inv_count - mean * mean, 0.f);
    out[(b * G + g) * 2 + 0] = mean;
    out[(b * G + g) * 2 + 1] = rsqrtf(var + eps);
  } else {
    out[(b * G + g) * 2 + 0] = s1 * inv_count;
    out[(b * G + g) * 2 + 1] = s2 * inv_count;
  }
}

// pass 2.  MODE 0: y = act(xhat*gamma+beta).  MODE 1: dx = rstd*(dxhat - m1 - xhat*m2)
template <int MODE>
__global__ __launch_bounds__(256) void gn_apply_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                       const bf16_t* __restrict__ gamma,
                                                       const bf16_t* __restrict__ beta,
                                                       const float* __restrict__ stats,
                                                       const float* __restrict__ dstats, int HW, int C, int silu,
                                                       const bf16_t* __restrict__ dres, bf16_t* __restrict__ out) {
  const int b = blockIdx.y;
  const int cols = C >> 3;
  const int cpg = C / G;
  const long nchunk = (long)HW * cols;
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < nchunk; id += (long)gridDim.x * blockDim.x) {
    const int col = (int)(id % cols);
    const long off = (long)b * HW * C + id * 8;
    float xv[8], ga[8], be[8], o[8];
    unpack8(*reinterpret_cast<const uint4*>(x + off), xv);
    unpack8(*reinterpret_cast<const uint4*>(gamma + col * 8), ga);
    unpack8(*reinterpret_cast<const uint4*>(beta + col * 8), be);
    float dv[8];
    if (MODE == 1) unpack8(*reinterpret_cast<const uint4*>(dy + off), dv);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int g = (col * 8 + j) / cpg;
      const float mu = stats[(b * G + g) * 2 + 0], rs = stats[(b * G + g) * 2 + 1];
      const float xh = (xv[j] - mu) * rs;
      const float z = xh * ga[j] + be[j];
      if (MODE == 0) {
        o[j] = silu ? z * sigmoidf_(z) : z;
      } else {
        float d = dv[j];
        if (silu) {
          const float s = sigmoidf_(z);
          d *= s * (1.f + z * (1.f - s));
        }
        d *= ga[j];
        o[j] = rs * (d - dstats[(b * G + g) * 2 + 0] - xh * dstats[(b * G + g) * 2 + 1]);
      }
    }
    if (MODE == 1 && dres != nullptr) {  // gradient of the residual / shortcut branch that shares this input
      float rsd[8];
      unpack8(*reinterpret_cast<const uint4*>(dres + off), rsd);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] += rsd[j];
    }
    *reinterpret_cast<uint4*>(out + off) = pack8(o);
  }
}

// pass 1, coalesced form for maps above 32x32 (round 3, session 5).  gn_fused_kernel<MODE, 1> reads a (sample, group, piece) as 4-byte
// accesses to 20-byte pixel segments (10 channels per group at 320): every 128-byte line is fetched by the workgroups of the ~6 groups
// that share it, and the pass measured 14 us for 21 MB (8 x 64x64 x 320) -- 1.5 TB/s.  Here a workgroup owns a slab of WHOLE pixel rows
// (the geometry of gn_apply2_kernel: one 16-byte chunk column per thread, GNS_U rows in flight), keeps per-channel fp32 sums in
// registers, folds them into the (at most two) groups its 8 channels belong to, and 64 threads add the workgroup's contributions in a
// fixed order: piece sums `part` [B][nslab][G][2] (slab-major: gn_apply2_kernel reads a piece's 64 sums as two cache lines).  Deterministic, no atomics, stateless.
// MODE 0: sum x, sum x^2.  MODE 1: sum dxhat, sum dxhat * xhat (the formulas of gn_apply2_kernel<1>).
constexpr int GNS_U = 4;
template <int MODE>
__global__ void gn_rowstats_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy, const bf16_t* __restrict__ gamma,
                                   const bf16_t* __restrict__ beta, const float* __restrict__ stats, int HW, int C,
                                   int rows_per_slab, int silu, float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) float sp[];   // [blockDim.x][4]: (g0 s1, g0 s2, g1 s1, g1 s2) of every thread
  const int cols = C >> 3, cpg = C / G;
  const int rp = blockDim.x / cols;
  const int col = threadIdx.x % cols, rr = threadIdx.x / cols;
  const int b = blockIdx.y, slab = blockIdx.x, nslab = gridDim.x;
  const int c0 = col * 8;
  const int g0 = c0 / cpg, g1 = (c0 + 7) / cpg;
  const int split = (g0 + 1) * cpg - c0;   // channels j < split belong to g0, the rest to g1
  float ga[8], be[8], mu2[2] = {0.f, 0.f}, rs2[2] = {0.f, 0.f};
  if (MODE == 1) {
    unpack8(*reinterpret_cast<const uint4*>(gamma + c0), ga);
    unpack8(*reinterpret_cast<const uint4*>(beta + c0), be);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int g = k ? g1 : g0;
      mu2[k] = stats[(b * G + g) * 2 + 0];
      rs2[k] = stats[(b * G + g) * 2 + 1];
    }
  }
  float a1[8], a2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) a1[j] = a2[j] = 0.f;
  const int r0 = slab * rows_per_slab, r1 = min(HW, r0 + rows_per_slab);
  for (int r = r0 + rr; r < r1; r += rp * GNS_U) {
    uint4 xw[GNS_U], dw[GNS_U];
#pragma unroll
    for (int u = 0; u < GNS_U; ++u) {
      const int ru = min(r + u * rp, r1 - 1);   // rows past the slab re-read its last row (dropped below)
      const long off = ((long)b * HW + ru) * C + c0;
      xw[u] = *reinterpret_cast<const uint4*>(x + off);
      if (MODE == 1) dw[u] = *reinterpret_cast<const uint4*>(dy + off);
    }
#pragma unroll
    for (int u = 0; u < GNS_U; ++u) {
      if (r + u * rp >= r1) continue;
      float xv[8], dv[8];
      unpack8(xw[u], xv);
      if (MODE == 1) unpack8(dw[u], dv);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (MODE == 0) {
          a1[j] += xv[j];
          a2[j] += xv[j] * xv[j];
        } else {
          const int k = j < split ? 0 : 1;
          const float xh = (xv[j] - mu2[k]) * rs2[k];
          float d = dv[j];
          if (silu) {
            const float z = xh * ga[j] + be[j];
            const float sg = sigmoidf_(z);
            d *= sg * (1.f + z * (1.f - sg));
          }
          d *= ga[j];
          a1[j] += d;
          a2[j] += d * xh;
        }
      }
    }
  }
  float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const bool lo = j < split;
    s[0] += lo ? a1[j] : 0.f;
    s[1] += lo ? a2[j] : 0.f;
    s[2] += lo ? 0.f : a1[j];
    s[3] += lo ? 0.f : a2[j];
  }
  *reinterpret_cast<float4*>(sp + threadIdx.x * 4) = make_float4(s[0], s[1], s[2], s[3]);
  __syncthreads();
  if (threadIdx.x < 2 * G) {   // fixed-order sum of the workgroup's contributions to (group, which)
    const int g = threadIdx.x >> 1, which = threadIdx.x & 1;
    const int c_lo = (g * cpg) >> 3, c_hi = ((g + 1) * cpg - 1) >> 3;
    float acc = 0.f;
    for (int q = 0; q < rp; ++q)
      for (int c = c_lo; c <= c_hi; ++c) {
        const int k = ((c * 8) / cpg == g) ? 0 : 1;   // the column starts inside g, or inside g - 1 and ends in g
        acc += sp[(q * cols + c) * 4 + k * 2 + which];
      }
    part[(((long)b * nslab + slab) * G + g) * 2 + which] = acc;   // slab-major: a piece's 64 sums are two cache lines
  }
}

// pass 2, second form (maps above 32x32).  A thread owns ONE 16-byte chunk column, so gamma / beta and the statistics of the (at
// most two) groups its 8 channels belong to are fetched once instead of 2-4 scalar loads per element, and it keeps GNA_U pixel
// rows in flight (the first form issues one chunk per thread and retires: at 21-42 MB per launch the kernel spent its time ramping
// wavefronts up and down).  Whole pixel rows are read and written contiguously.  `part` != null: the statistics come from the
// piece sums of gn_fused_kernel<0, 1> (summed in the same fixed order as its PHASE 2), and are written to `stats_out` for backward.
constexpr int GNA_U = 4;
template <int MODE>
__global__ void gn_apply2_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy, const bf16_t* __restrict__ gamma,
                                 const bf16_t* __restrict__ beta, const float* __restrict__ stats,
                                 const float* __restrict__ dstats, const float* __restrict__ part, int nsplit, int slab_major,
                                 float inv_count,
                                 float eps, float* __restrict__ stats_out, int HW, int C, int rows_per_slab, int silu,
                                 const bf16_t* __restrict__ dres, bf16_t* __restrict__ out) {
  const int cols = C >> 3, cpg = C / G;
  const int rp = blockDim.x / cols;
  const int col = threadIdx.x % cols, rr = threadIdx.x / cols;
  const int b = blockIdx.y, slab = blockIdx.x;
  const int c0 = col * 8;
  const int g0 = c0 / cpg, g1 = (c0 + 7) / cpg;   // cpg >= 8 is not required: with cpg < 8 the callers use the first form
  const int split = (g0 + 1) * cpg - c0;          // channels j < split belong to g0, the rest to g1
  // Piece sums (gn_fused_kernel<MODE, 1>: <= 8 pieces; gn_rowstats_kernel: up to 128 slabs): the workgroup adds them ONCE,
  // cooperatively -- (group, which) = t & 63, pieces t >> 6, t >> 6 + nq, ... with eight loads in flight, then the nq partial sums in
  // order -- instead of every thread walking the pieces of its two groups one L2 round trip at a time.  Fixed order: deterministic.
  __shared__ float gn_red[8][2 * G];
  __shared__ float gn_tot[2 * G];
  if (part != nullptr) {
    const int t = threadIdx.x;
    int nq = blockDim.x >> 6;
    nq = nq > 8 ? 8 : (nq < 1 ? 1 : nq);
    if (t < nq * 64 && t < (int)blockDim.x) {
      const int gw = t & 63, q0 = t >> 6;
      // group-major [B][G][nsplit][2] (gn_fused_kernel<MODE, 1>) or slab-major [B][nsplit][G][2] (gn_rowstats_kernel)
      const float* src = slab_major ? part + (long)b * nsplit * (2 * G) + gw : part + ((long)b * G + (gw >> 1)) * nsplit * 2 + (gw & 1);
      const long qs = slab_major ? 2 * G : 2;
      float acc = 0.f;
      for (int q = q0; q < nsplit; q += nq * 8) {
        float u[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) u[i] = src[(long)min(q + i * nq, nsplit - 1) * qs];
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (q + i * nq < nsplit) acc += u[i];
      }
      gn_red[q0][gw] = acc;
    }
    __syncthreads();
    if (t < 2 * G) {
      float a = 0.f;
      for (int q = 0; q < nq; ++q) a += gn_red[q][t];
      gn_tot[t] = a;
    }
    __syncthreads();
  }
  if (rr >= rp) return;
  float ga[8], be[8];
  unpack8(*reinterpret_cast<const uint4*>(gamma + c0), ga);
  unpack8(*reinterpret_cast<const uint4*>(beta + c0), be);
  float mu2[2], rs2[2], m12[2] = {0.f, 0.f}, m22[2] = {0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int g = k ? g1 : g0;
    float a0 = 0.f, a1 = 0.f;
    if (part != nullptr) {
      a0 = gn_tot[2 * g + 0];
      a1 = gn_tot[2 * g + 1];
    }
    if (MODE == 0 && part != nullptr) {
      mu2[k] = a0 * inv_count;
      rs2[k] = rsqrtf(fmaxf(a1 * inv_count - mu2[k] * mu2[k], 0.f) + eps);
      if (slab == 0 && rr == 0 && ((g * cpg) >> 3) == col && (k == 0 || g1 != g0)) {
        stats_out[(b * G + g) * 2 + 0] = mu2[k];
        stats_out[(b * G + g) * 2 + 1] = rs2[k];
      }
    } else {
      mu2[k] = stats[(b * G + g) * 2 + 0];
      rs2[k] = stats[(b * G + g) * 2 + 1];
    }
    if (MODE == 1) {   // mean(dxhat), mean(dxhat * xhat): from the piece sums, or finalized by gn_finalize_kernel<1>
      m12[k] = part != nullptr ? a0 * inv_count : dstats[(b * G + g) * 2 + 0];
      m22[k] = part != nullptr ? a1 * inv_count : dstats[(b * G + g) * 2 + 1];
    }
  }
  const int r0 = slab * rows_per_slab, r1 = min(HW, r0 + rows_per_slab);
  for (int r = r0 + rr; r < r1; r += rp * GNA_U) {
    uint4 xw[GNA_U], dw[GNA_U], rw[GNA_U];
#pragma unroll
    for (int u = 0; u < GNA_U; ++u) {
      const int ru = min(r + u * rp, r1 - 1);   // rows past the slab re-read its last row (dropped below)
      const long off = ((long)b * HW + ru) * C + c0;
      xw[u] = *reinterpret_cast<const uint4*>(x + off);
      if (MODE == 1) {
        dw[u] = *reinterpret_cast<const uint4*>(dy + off);
        if (dres != nullptr) rw[u] = *reinterpret_cast<const uint4*>(dres + off);
      }
    }
#pragma unroll
    for (int u = 0; u < GNA_U; ++u) {
      if (r + u * rp >= r1) continue;
      const long off = ((long)b * HW + r + u * rp) * C + c0;
      float xv[8], dv[8], o[8];
      unpack8(xw[u], xv);
      if (MODE == 1) unpack8(dw[u], dv);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = j < split ? 0 : 1;
        const float mu = mu2[k], rs = rs2[k];
        const float xh = (xv[j] - mu) * rs;
        const float z = xh * ga[j] + be[j];
        if (MODE == 0) {
          o[j] = silu ? z * sigmoidf_(z) : z;
        } else {
          float d = dv[j];
          if (silu) {
            const float sg = sigmoidf_(z);
            d *= sg * (1.f + z * (1.f - sg));
          }
          d *= ga[j];
          o[j] = rs * (d - m12[k] - xh * m22[k]);
        }
      }
      if (MODE == 1 && dres != nullptr) {
        float rsd[8];
        unpack8(rw[u], rsd);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] += rsd[j];
      }
      *reinterpret_cast<uint4*>(out + off) = pack8(o);
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Fused GroupNorm: ONE launch, one workgroup per (sample, group).  The workgroup streams its HW x (C/32) slice twice --
// pass 1 statistics, pass 2 normalise (+SiLU) -- and the second read is served by the XCD's L2 (a slice is 82 KB at
// 64x64x320, 245 KB at 64x64x960), so HBM sees one read and one write and nothing but a workgroup barrier separates the
// passes (the 3-kernel form drains the chip twice per GroupNorm, 180 times a step).  Accesses are 4-byte (two
// channels): group boundaries are only 4-byte aligned (10 / 20 / 30 / 60 channels per group in the SD-1.5 U-Net);
// consecutive lanes walk the dwords of a pixel, then the next pixel.  Block mapping: hardware block b runs on XCD b % 8;
// groups 4x .. 4x+3 of every sample go to XCD x so that the cache lines they share are fetched by one L2.
// Requires an even number of channels per group (else the callers use the 3-kernel path).
// MODE 0: y = act(xhat*gamma+beta), writes stats.  MODE 1: dx = rstd*(dxhat - mean(dxhat) - xhat*mean(dxhat*xhat)) (+dres)
// ---------------------------------------------------------------------------------------------------
constexpr int GNF_THREADS = 1024;
constexpr int GNF_U = 8;  // pixels in flight per thread

// PHASE 0: the whole (sample, group) in one workgroup (maps up to 32x32).  Larger maps split the pixel range over `nsplit`
// workgroups and two launches: PHASE 1 writes each piece's raw sums to `part` [B][32][nsplit][2], PHASE 2 adds the pieces in
// a fixed order (every workgroup redundantly: nsplit <= 8 pairs), then normalises its own piece -- one chip-wide drain
// instead of two, no slab scratch traffic, and the pieces of a group stay on one XCD.
// SRC = 1 (round 6, PHASE 0 only): the tensor a split-K GEMM / conv produced is still its fp32 partial slabs [splits][B*HW][C] -- the
// GroupNorm's first pass sums them and applies the GEMM's bf16 epilogue (bias, per-sample row bias, residual: the arithmetic of
// splitk_finalize_kernel, operation for operation), i.e. the finalize launch between a split-K convolution and the GroupNorm behind it
// disappears.  MODE 0: the finished tensor is x; the pass also WRITES it (`xout`: backward and residual branches read it) and the
// second pass reads back what the same thread stored.  MODE 1: the finished tensor is dy (a backward-data convolution's output,
// no epilogue terms); it is consumed here and never reaches HBM -- both passes sum the slabs.
struct GnSlabSrc {
  const float* slabs;
  int splits;
  long slab_stride;          // B * HW * C floats
  const bf16_t* bias;        // [C] or null
  const bf16_t* rowbias;     // [B][rowbias_ld] or null
  long rowbias_ld;
  const bf16_t* residual;    // [B][HW][C] or null
  bf16_t* xout;              // MODE 0: the finished tensor [B][HW][C]
};

template <int MODE, int PHASE, int SRC = 0>
__global__ __launch_bounds__(GNF_THREADS) void gn_fused_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                               const bf16_t* __restrict__ gamma,
                                                               const bf16_t* __restrict__ beta, float* __restrict__ stats,
                                                               int HW, int C, float eps, int silu,
                                                               const bf16_t* __restrict__ dres, bf16_t* __restrict__ out,
                                                               int nsplit, float* __restrict__ part, const GnSlabSrc ss) {
  static_assert(SRC == 0 || PHASE == 0, "the slab source exists for the one-launch form only");
  __shared__ float red[2][GNF_THREADS / 64];
  __shared__ float bc[2];
  const int bid = blockIdx.x;
  // hardware block id -> XCD bid % 8, which hosts groups 4x..4x+3 of every sample and all their pixel pieces
  const int rest = bid >> 3, jg = rest & 3, sp = (rest >> 2) % nsplit, b = (rest >> 2) / nsplit;
  const int g = ((bid & 7) << 2) | jg;
  const int ppw = (HW + nsplit - 1) / nsplit;           // pixels per piece
  const int p_lo = sp * ppw, p_hi = min(HW, p_lo + ppw);
  const int cpg = C / G, D = cpg >> 1;        // dwords per pixel of this group
  const int active = (GNF_THREADS / D) * D;   // each thread keeps ONE channel pair
  const int tid = threadIdx.x;
  const bool live = tid < active;
  const int dw = tid % D, p0 = tid / D, pstep = active / D;
  const int ch = g * cpg + 2 * dw;
  const long base = (long)b * HW * C + ch;
  float ga0 = 0.f, ga1 = 0.f, be0 = 0.f, be1 = 0.f;
  {
    const uint32_t gw = *reinterpret_cast<const uint32_t*>(gamma + ch), bw = *reinterpret_cast<const uint32_t*>(beta + ch);
    ga0 = bf16lo(gw); ga1 = bf16hi(gw); be0 = bf16lo(bw); be1 = bf16hi(bw);
  }
  float mu, rs;
  if (MODE == 1) {
    mu = stats[(b * G + g) * 2 + 0];
    rs = stats[(b * G + g) * 2 + 1];
  }
  auto dxhat = [&](float xh, float d, float ga, float be) {
    if (silu) {
      const float z = xh * ga + be;
      const float sg = sigmoidf_(z);
      d *= sg * (1.f + z * (1.f - sg));
    }
    return d * ga;
  };
  // SRC = 1: the channel pair of GNF_U pixels out of the split-K slabs, finished as splitk_finalize_kernel finishes it (fp32 sum over
  // the splits in order, + bias in fp32, ONE rounding to bf16, then the row bias and the residual as bf16 adds)
  // Memory-level parallelism: a thread owns only ~5-10 channel pairs of its (sample, group), each the sum of up to 16 slabs -- the pass
  // is pure latency unless many loads fly at once.  Per step: the SAME GNF_ZC splits of GNF_U / 2 pixels (16 eight-byte loads in flight),
  // splits past the end clamped to the last one and masked out of the sum; sums are taken in split order, as the finalize kernel takes them.
  constexpr int GNF_ZC = 4, GNF_US = GNF_U / 2;
  auto from_slabs = [&](int p, int p_end, uint32_t (&w)[GNF_U]) {
    uint32_t rbw = 0u;
    float bi0 = 0.f, bi1 = 0.f;
    if (MODE == 0) {
      if (ss.bias != nullptr) {
        const uint32_t t = *reinterpret_cast<const uint32_t*>(ss.bias + ch);
        bi0 = bf16lo(t), bi1 = bf16hi(t);
      }
      if (ss.rowbias != nullptr) rbw = *reinterpret_cast<const uint32_t*>(ss.rowbias + (long)b * ss.rowbias_ld + ch);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {        // two halves of the GNF_U pixels
      if (p + h * GNF_US * pstep >= p_end) {   // (thread-uniform over u: nothing of this half is live)
#pragma unroll
        for (int u = 0; u < GNF_US; ++u) w[h * GNF_US + u] = 0u;
        continue;
      }
      float2 acc[GNF_US];
      long po[GNF_US];
#pragma unroll
      for (int u = 0; u < GNF_US; ++u) {
        acc[u] = make_float2(0.f, 0.f);
        po[u] = base + (long)min(p + (h * GNF_US + u) * pstep, p_end - 1) * C;
      }
      uint32_t rsw[GNF_US];
      if (MODE == 0 && ss.residual != nullptr) {
#pragma unroll
        for (int u = 0; u < GNF_US; ++u) rsw[u] = *reinterpret_cast<const uint32_t*>(ss.residual + po[u]);
      }
      for (int z0 = 0; z0 < ss.splits; z0 += GNF_ZC) {
        float2 t[GNF_ZC][GNF_US];
#pragma unroll
        for (int zc = 0; zc < GNF_ZC; ++zc) {
          const float* sl = ss.slabs + (long)min(z0 + zc, ss.splits - 1) * ss.slab_stride;
#pragma unroll
          for (int u = 0; u < GNF_US; ++u) t[zc][u] = *reinterpret_cast<const float2*>(sl + po[u]);
        }
#pragma unroll
        for (int zc = 0; zc < GNF_ZC; ++zc) {
          if (z0 + zc < ss.splits) {
#pragma unroll
            for (int u = 0; u < GNF_US; ++u) acc[u].x += t[zc][u].x, acc[u].y += t[zc][u].y;
          }
        }
      }
#pragma unroll
      for (int u = 0; u < GNF_US; ++u) {
        uint32_t v = pack_bf16x2(acc[u].x + bi0, acc[u].y + bi1);
        if (MODE == 0 && ss.rowbias != nullptr) v = pack_bf16x2(bf16lo(v) + bf16lo(rbw), bf16hi(v) + bf16hi(rbw));
        if (MODE == 0 && ss.residual != nullptr) v = pack_bf16x2(bf16lo(v) + bf16lo(rsw[u]), bf16hi(v) + bf16hi(rsw[u]));
        w[h * GNF_US + u] = v;
      }
    }
  };
  // ---- pass 1.  GNF_U pixels per thread are loaded before any is used: a thread has only HW*D/1020 (~20) dwords to
  // fetch per pass, so the pass is latency-bound unless they are all in flight
  float s1 = 0.f, s2 = 0.f;
  // SRC = 1: when a thread's whole share is ONE step of GNF_U pixels (maps up to ~16 x 16 at 1280 channels, 32 x 32 at 320), the finished
  // values stay in registers for the second pass -- no second sum over the slabs, no read-back of the tensor just written
  const bool single = SRC == 1 && (p_hi - p_lo) <= pstep * GNF_U;
  uint32_t keep_x[GNF_U], keep_d[GNF_U];
  if (live && PHASE != 2) {
    for (int p = p_lo + p0; p < p_hi; p += pstep * GNF_U) {
      uint32_t xw[GNF_U], dwv[GNF_U];
      if (SRC == 1 && MODE == 0) {
        from_slabs(p, p_hi, xw);
#pragma unroll
        for (int u = 0; u < GNF_U; ++u)
          if (p + u * pstep < p_hi) *reinterpret_cast<uint32_t*>(ss.xout + base + (long)(p + u * pstep) * C) = xw[u];
      }
      if (SRC == 1 && MODE == 1) from_slabs(p, p_hi, dwv);
#pragma unroll
      for (int u = 0; u < GNF_U; ++u) {
        const int pp = min(p + u * pstep, p_hi - 1);
        if (!(SRC == 1 && MODE == 0)) xw[u] = *reinterpret_cast<const uint32_t*>(x + base + (long)pp * C);
        if (MODE == 1 && SRC == 0) dwv[u] = *reinterpret_cast<const uint32_t*>(dy + base + (long)pp * C);
      }
      if (SRC == 1) {
#pragma unroll
        for (int u = 0; u < GNF_U; ++u) {
          keep_x[u] = xw[u];
          if (MODE == 1) keep_d[u] = dwv[u];
        }
      }
#pragma unroll
      for (int u = 0; u < GNF_U; ++u) {
        if (p + u * pstep >= p_hi) continue;
        const float x0 = bf16lo(xw[u]), x1 = bf16hi(xw[u]);
        if (MODE == 0) {
          s1 += x0 + x1;
          s2 += x0 * x0 + x1 * x1;
        } else {
          const float h0 = (x0 - mu) * rs, h1 = (x1 - mu) * rs;
          const float d0 = dxhat(h0, bf16lo(dwv[u]), ga0, be0), d1 = dxhat(h1, bf16hi(dwv[u]), ga1, be1);
          s1 += d0 + d1;
          s2 += d0 * h0 + d1 * h1;
        }
      }
    }
  }
  if (PHASE != 2) {
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if ((tid & 63) == 0) {
      red[0][tid >> 6] = s1;
      red[1][tid >> 6] = s2;
    }
    __syncthreads();
    if (tid < 2) {  // fixed-order sum over the 16 wavefronts: deterministic
      float a = 0.f;
      for (int w = 0; w < GNF_THREADS / 64; ++w) a += red[tid][w];
      bc[tid] = a;
      if (PHASE == 1) part[(((long)b * G + g) * nsplit + sp) * 2 + tid] = a;
    }
    if (PHASE == 1) return;
  } else {
    if (tid < 2) {
      float a = 0.f;
      for (int q = 0; q < nsplit; ++q) a += part[(((long)b * G + g) * nsplit + q) * 2 + tid];
      bc[tid] = a;
    }
  }
  __syncthreads();
  const float inv = 1.f / ((float)HW * cpg);
  float m1, m2;
  if (MODE == 0) {
    mu = bc[0] * inv;
    rs = rsqrtf(fmaxf(bc[1] * inv - mu * mu, 0.f) + eps);
    if (tid == 0 && sp == 0) {
      stats[(b * G + g) * 2 + 0] = mu;
      stats[(b * G + g) * 2 + 1] = rs;
    }
  } else {
    m1 = bc[0] * inv;
    m2 = bc[1] * inv;
  }
  // ---- pass 2 (re-read from L2)
  if (!live) return;
  for (int p = p_lo + p0; p < p_hi; p += pstep * GNF_U) {
    uint32_t xw[GNF_U], dwv[GNF_U], rw[GNF_U];
    if (SRC == 1 && MODE == 1 && !single) from_slabs(p, p_hi, dwv);
    // (SRC = 1, MODE 0: x == ss.xout, written by THIS thread in pass 1 -- same pixel / channel-pair mapping in both passes)
    const bf16_t* xs = (SRC == 1 && MODE == 0) ? ss.xout : x;
#pragma unroll
    for (int u = 0; u < GNF_U; ++u) {
      const long o = base + (long)min(p + u * pstep, p_hi - 1) * C;
      if (SRC == 1 && single) {
        xw[u] = keep_x[u];
        if (MODE == 1) dwv[u] = keep_d[u];
      } else {
        xw[u] = *reinterpret_cast<const uint32_t*>(xs + o);
      }
      if (MODE == 1) {
        if (SRC == 0) dwv[u] = *reinterpret_cast<const uint32_t*>(dy + o);
        rw[u] = dres != nullptr ? *reinterpret_cast<const uint32_t*>(dres + o) : 0u;
      }
    }
#pragma unroll
    for (int u = 0; u < GNF_U; ++u) {
      if (p + u * pstep >= p_hi) continue;
      const long o = base + (long)(p + u * pstep) * C;
      const float h0 = (bf16lo(xw[u]) - mu) * rs, h1 = (bf16hi(xw[u]) - mu) * rs;
      float r0, r1;
      if (MODE == 0) {
        const float z0 = h0 * ga0 + be0, z1 = h1 * ga1 + be1;
        r0 = silu ? z0 * sigmoidf_(z0) : z0;
        r1 = silu ? z1 * sigmoidf_(z1) : z1;
      } else {
        r0 = rs * (dxhat(h0, bf16lo(dwv[u]), ga0, be0) - m1 - h0 * m2) + bf16lo(rw[u]);
        r1 = rs * (dxhat(h1, bf16hi(dwv[u]), ga1, be1) - m1 - h1 * m2) + bf16hi(rw[u]);
      }
      *reinterpret_cast<uint32_t*>(out + o) = pack_bf16x2(r0, r1);
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// LayerNorm: one wavefront per token row, row kept in registers (C <= 64*8*MAXC).
// ---------------------------------------------------------------------------------------------------
constexpr int LN_MAXC = 3;  // up to 1536 channels

// R rows per wavefront, every row's loads issued before the first is consumed: with one row per wavefront the kernel has
// 32 waves x 640 B = 20 KB in flight per CU at 320 channels (5 MB on the chip: ~3 TB/s at ~1.7 us of latency, what it measured);
// two rows double that at the same occupancy now that the reductions no longer go through the LDS pipe.
template <int MODE, int R, int NT>
__global__ __launch_bounds__(256) void ln_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                 const bf16_t* __restrict__ gamma, const bf16_t* __restrict__ beta,
                                                 float* __restrict__ stats, long M, int C, float eps,
                                                 bf16_t* __restrict__ out) {
#pragma clang fp contract(off)   // every fused multiply-add below is an explicit fmaf: the same bits in every (R, NT) instance
  const int lane = threadIdx.x & 63;
  const long row0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
  if (row0 >= M) return;
  const int cols = C >> 3;
  uint4 xr[R][NT], dr[R][NT], rr[R][NT];
  float st0[R], st1[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const long row = min(row0 + r, M - 1);   // rows past the end re-read the last row (not stored)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int c = lane + t * 64;
      if (c < cols) {
        xr[r][t] = *reinterpret_cast<const uint4*>(x + row * C + c * 8);
        if (MODE == 1) {
          dr[r][t] = *reinterpret_cast<const uint4*>(dy + row * C + c * 8);
          if (beta != nullptr) rr[r][t] = *reinterpret_cast<const uint4*>(beta + row * C + c * 8);
        }
      }
    }
    if (MODE == 1) {
      st0[r] = stats[row * 2 + 0];
      st1[r] = stats[row * 2 + 1];
    }
  }
  uint4 gr[NT], br[NT];   // gamma (and beta) as loaded: unpacked where they are used (registers bound the rows in flight)
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int c = lane + t * 64;
    if (c < cols) {
      gr[t] = *reinterpret_cast<const uint4*>(gamma + c * 8);
      if (MODE == 0) br[t] = *reinterpret_cast<const uint4*>(beta + c * 8);
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const long row = row0 + r;
    if (row >= M) break;
    float xv[NT][8];
    float dv[NT][8];
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int c = lane + t * 64;
      if (c < cols) {
        unpack8(xr[r][t], xv[t]);
#pragma unroll
        for (int j = 0; j < 8; ++j) s += xv[t][j];
      }
    }
    float mean, rstd;
    if (MODE == 0) {
      mean = wave_sum(s) / C;
      float v = 0.f;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int c = lane + t * 64;
        if (c < cols) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float d = xv[t][j] - mean;
            v = fmaf(d, d, v);
          }
        }
      }
      rstd = rsqrtf(wave_sum(v) / C + eps);
      if (lane == 0) {
        stats[row * 2 + 0] = mean;
        stats[row * 2 + 1] = rstd;
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int c = lane + t * 64;
        if (c < cols) {
          float o[8], ga[8], be[8];
          unpack8(gr[t], ga);
          unpack8(br[t], be);
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] = fmaf((xv[t][j] - mean) * rstd, ga[j], be[j]);
          *reinterpret_cast<uint4*>(out + row * C + c * 8) = pack8(o);
        }
      }
    } else {
      mean = st0[r];
      rstd = st1[r];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int c = lane + t * 64;
        if (c < cols) {
          float ga[8];
          unpack8(gr[t], ga);
          unpack8(dr[r][t], dv[t]);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            dv[t][j] *= ga[j];
            xv[t][j] = (xv[t][j] - mean) * rstd;
            s1 += dv[t][j];
            s2 = fmaf(dv[t][j], xv[t][j], s2);
          }
        }
      }
      s1 = wave_sum(s1) / C;
      s2 = wave_sum(s2) / C;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int c = lane + t * 64;
        if (c < cols) {
          float o[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] = rstd * fmaf(-xv[t][j], s2, dv[t][j] - s1);
          if (beta != nullptr) {  // backward: `beta` carries the gradient of the residual branch (same shape as x)
            float rsd[8];
            unpack8(rr[r][t], rsd);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] += rsd[j];
          }
          *reinterpret_cast<uint4*>(out + row * C + c * 8) = pack8(o);
        }
      }
    }
  }
}

template <int MODE>
void ln_launch(bool two_rows, const bf16_t* x, const bf16_t* dy, const bf16_t* gamma, const bf16_t* beta, float* stats, long M, int C,
               float eps, bf16_t* out, hipStream_t stream) {
  const int nt = (C / 8 + 63) / 64;   // 16-byte chunk slots per lane: 1 up to 512 channels, 2 up to 1024, 3 up to 1536
#define AQL_LN(R, NT)                                                                                                         \
  hipLaunchKernelGGL((ln_kernel<MODE, R, NT>), dim3((unsigned)((M + 4 * R - 1) / (4 * R))), dim3(256), 0, stream, x, dy, gamma, beta, \
                     stats, M, C, eps, out)
  if (two_rows) {
    if (nt == 1) AQL_LN(2, 1);
    else if (nt == 2) AQL_LN(2, 2);
    else AQL_LN(2, 3);
  } else {
    if (nt == 1) AQL_LN(1, 1);
    else if (nt == 2) AQL_LN(1, 2);
    else AQL_LN(1, 3);
  }
#undef AQL_LN
}

inline int gn_geometry(int HW, int C, int* threads, int* rows_per_slab) {
  const int cols = C / 8;
  int rp = 512 / cols;
  if (rp < 1) rp = 1;
  if (rp > HW) rp = HW;
  *threads = cols * rp;
  // ~32 pixel rows per slab; 64 up to 320 channels (12 rows in flight per workgroup: two rounds of GNA_U loads instead of one short
  // one -- 64x64x320: 21.3 -> 18.3 us forward at 8 samples, 14.3 -> 12.8 at 4; neutral or slower at 640 / 960).  Tuning hook.
  static const int slab_env = AQL_TUNE_INT("AQL_GN_SLAB_ROWS", 0);
  const int slab_rows = slab_env > 0 ? slab_env : (C <= 320 ? 64 : 32);
  int nslab = (HW + slab_rows - 1) / slab_rows;
  if (nslab > 128) nslab = 128;
  if (nslab < 1) nslab = 1;
  *rows_per_slab = (HW + nslab - 1) / nslab;
  return (HW + *rows_per_slab - 1) / *rows_per_slab;
}

// Fused single-launch form: needs an even channel count per group (4-byte accesses).  Measured per shape inside a HIP
// graph (tools/tune_gn.py, B=4): 2-4x faster than the 3-kernel form up to 32x32 maps (8.1 vs 14.7 us at 320 ch, 13.6 vs
// 23.4 at 1280 ch; backward 4.7 vs 22.3 us at 8x8x1280), slower at 64x64 (33.5 vs 26.5 us at 320 ch: a group's 20-byte
// pixel segments touch 6x their bytes in cache lines and 128 workgroups cannot hide it), so large maps keep the slab
// form.  AQL_GN_FUSED=0 forces the 3-kernel path, =2 forces the fused one (tuning).
// 32x32 maps, backward, <= 960 channels: the split form (two launches) is 2-4 us faster than the one-launch form (B=4,
// tools/tune_gn.py: 16.1 -> 12.6 us at 640 channels, 18.9 -> 15.3 at 960, 11.8 -> 9.9 at 320; 1280 / 1920 channels and every
// forward shape are within 1 us or slower).  AQL_GN_FUSED_MAXHW forces the threshold for all shapes (tuning hook).
inline bool gn_use_fused(int C, int HW, bool bwd = false) {
  static const int en = AQL_TUNE_INT("AQL_GN_FUSED", 1);
  if (!en || (C / G) % 2 != 0 || (C / G) / 2 > GNF_THREADS) return false;
  static const int max_hw = AQL_TUNE_INT("AQL_GN_FUSED_MAXHW", 0);
  if (en == 2) return true;
  if (max_hw > 0) return HW <= max_hw;
  if (bwd && HW > 512 && C <= 960 && C / G >= 8) return false;
  return HW <= 2048;
}

// Coalesced statistics pass (gn_rowstats_kernel) + gn_apply2_kernel for every map the one-launch form does not take, forward and
// backward: AQL_GN_ROWSTATS=0 restores the per-(sample, group, piece) first pass (gn_split), =n sets the slab count of the pass.
// Measured (tools/tune_gn.py, profiles/r03_gn_rowstats.txt): 64x64 forward at 8 samples 25.6 -> 21.1 us (320 channels), 32.5 -> 28.4 (640),
// 40.3 -> 36.6 (960); backward at 4 samples 23.7 -> 17.3, 30.4 -> 26.1, 38.5 -> 34.4; 32x32 backward: 4 samples 10.3 -> 11.5 (kept on the
// old form), 8 samples 15.1 -> 12.1.  Slabs: ~512 workgroups in all (128 slabs at 4 samples, 64 at 8).
inline int gn_rowstats_slabs(int B, int HW, int C, int nslab_apply) {
  static const int en = AQL_TUNE_INT("AQL_GN_ROWSTATS", 1);
  if (en == 0 || (C / G) < 8 || C / 8 > 512) return 0;
  if (en == 1 && HW <= 1024 && B < 8) return 0;
  int ns = en > 1 ? en : 512 / (B > 0 ? B : 1);
  if (ns < 32) ns = 32;
  if (ns > 128) ns = 128;
  if (ns > nslab_apply && en == 1) ns = nslab_apply;
  if (ns > HW) ns = HW;
  return ns;
}

// larger maps, forward only: two launches of the same kernel with the pixel range of every (sample, group) cut into pieces
// (64x64, B=4, tools/tune_gn.py: 22.3 vs 26.5 us at 320 channels, 25.4 vs 34.9 at 640, 33.5 vs 41.7 at 960 against the slab
// form).  0 = use the slab form; AQL_GN_SPLIT=0 disables, =n forces n pieces
inline int gn_split(int C, int HW) {
  static const int en = AQL_TUNE_INT("AQL_GN_SPLIT", -1);
  if (en == 0 || (C / G) % 2 != 0 || (C / G) / 2 > GNF_THREADS || HW <= 512 || HW > 8192) return 0;   // callers ask only when the one-launch form is off
  if (en > 0) return en > 8 ? 8 : en;
  int ns = HW / 1024;
  if (ns < 2) ns = 2;
  return ns > 8 ? 8 : ns;
}

}  // namespace

// scratch: caller-owned fp32 workspace of at least aql_groupnorm_scratch_floats(B, HW) elements
extern "C" long aql_groupnorm_scratch_floats(int B, int HW) {
  (void)HW;
  return (long)B * 128 * G * 2 + (long)B * G * 2;
}

extern "C" int aql_groupnorm_silu_fwd(const bf16_t* x, int B, int HW, int C, const bf16_t* gamma, const bf16_t* beta,
                                      float eps, int silu, bf16_t* y, float* stats, float* scratch,
                                      hipStream_t stream) {
  AQL_CHECK_ARG(x && gamma && beta && y && stats && scratch, "aql_groupnorm_silu_fwd: null operand");
  AQL_CHECK_ARG(C % (8 * 1) == 0 && C % G == 0 && C / 8 <= 1024, "aql_groupnorm_silu_fwd: bad C=%d", C);
  if (gn_use_fused(C, HW)) {
    hipLaunchKernelGGL((gn_fused_kernel<0, 0>), dim3(B * G), dim3(GNF_THREADS), 0, stream, x, nullptr, gamma, beta, stats, HW,
                       C, eps, silu, nullptr, y, 1, nullptr, GnSlabSrc{});
    AQL_CHECK_LAUNCH("aql_groupnorm_silu_fwd");
    return AQL_OK;
  }
  static const int apply2 = AQL_TUNE_INT("AQL_GN_APPLY2", 1);   // A/B hook
  int threads, rps;
  const int nslab = gn_geometry(HW, C, &threads, &rps);
  const bool a2_ok = apply2 && (C / G) >= 8;
  if (const int nss = a2_ok ? gn_rowstats_slabs(B, HW, C, nslab) : 0) {
    const int rps_s = (HW + nss - 1) / nss, ns = (HW + rps_s - 1) / rps_s;
#ifdef AQL_EXPERIMENTS
    // timing bound only (WRONG results): inside a graph capture the statistics launch is left out -- the scratch then holds the finite
    // sums some other layer left there during the eager warm-up steps
    static const int skip_stats = getenv("AQL_EXP_GN_SKIPSTATS") ? atoi(getenv("AQL_EXP_GN_SKIPSTATS")) : 0;
    hipStreamCaptureStatus cst = hipStreamCaptureStatusNone;
    if (skip_stats) (void)hipStreamIsCapturing(stream, &cst);
    if (!(skip_stats && cst == hipStreamCaptureStatusActive))
#endif
    hipLaunchKernelGGL(gn_rowstats_kernel<0>, dim3(ns, B), dim3(threads), threads * 16, stream, x, nullptr, gamma, beta, nullptr, HW, C,
                       rps_s, silu, scratch);
    hipLaunchKernelGGL(gn_apply2_kernel<0>, dim3(nslab, B), dim3(threads), 0, stream, x, nullptr, gamma, beta, nullptr, nullptr,
                       scratch, ns, 1, 1.f / ((float)HW * (C / G)), eps, stats, HW, C, rps, silu, nullptr, y);
    AQL_CHECK_LAUNCH("aql_groupnorm_silu_fwd");
    return AQL_OK;
  }
  if (const int ns = gn_split(C, HW)) {
    hipLaunchKernelGGL((gn_fused_kernel<0, 1>), dim3(B * G * ns), dim3(GNF_THREADS), 0, stream, x, nullptr, gamma, beta, stats,
                       HW, C, eps, silu, nullptr, y, ns, scratch, GnSlabSrc{});
    if (a2_ok)   // coalesced second pass that sums the piece statistics itself
      hipLaunchKernelGGL(gn_apply2_kernel<0>, dim3(nslab, B), dim3(threads), 0, stream, x, nullptr, gamma, beta, nullptr, nullptr,
                         scratch, ns, 0, 1.f / ((float)HW * (C / G)), eps, stats, HW, C, rps, silu, nullptr, y);
    else
      hipLaunchKernelGGL((gn_fused_kernel<0, 2>), dim3(B * G * ns), dim3(GNF_THREADS), 0, stream, x, nullptr, gamma, beta, stats,
                         HW, C, eps, silu, nullptr, y, ns, scratch, GnSlabSrc{});
    AQL_CHECK_LAUNCH("aql_groupnorm_silu_fwd");
    return AQL_OK;
  }
  hipLaunchKernelGGL(gn_partial_kernel<0>, dim3(nslab, B), dim3(threads), threads * 64, stream, x, nullptr, gamma, beta, nullptr,
                     HW, C, rps, silu, scratch);
  hipLaunchKernelGGL(gn_finalize_kernel<0>, dim3(B), dim3(256), 0, stream, scratch, nslab,
                     1.f / ((float)HW * (C / G)), eps, stats);
  if (a2_ok) {
    hipLaunchKernelGGL(gn_apply2_kernel<0>, dim3(nslab, B), dim3(threads), 0, stream, x, nullptr, gamma, beta, stats, nullptr,
                       nullptr, 0, 0, 0.f, 0.f, nullptr, HW, C, rps, silu, nullptr, y);
    AQL_CHECK_LAUNCH("aql_groupnorm_silu_fwd");
    return AQL_OK;
  }
  const long nchunk = (long)HW * (C / 8);
  int blocks = (int)((nchunk + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(gn_apply_kernel<0>, dim3(blocks, B), dim3(256), 0, stream, x, nullptr, gamma, beta, stats, nullptr,
                     HW, C, silu, nullptr, y);
  AQL_CHECK_LAUNCH("aql_groupnorm_silu_fwd");
  return AQL_OK;
}

extern "C" int aql_groupnorm_silu_bwd(const bf16_t* x, const bf16_t* dy, int B, int HW, int C, const bf16_t* gamma,
                                      const bf16_t* beta, int silu, const float* stats, const bf16_t* dres, bf16_t* dx,
                                      float* scratch, hipStream_t stream) {
  AQL_CHECK_ARG(x && dy && gamma && beta && dx && stats && scratch, "aql_groupnorm_silu_bwd: null operand");
  AQL_CHECK_ARG(C % 8 == 0 && C % G == 0 && C / 8 <= 1024, "aql_groupnorm_silu_bwd: bad C=%d", C);
  if (gn_use_fused(C, HW, true)) {
    hipLaunchKernelGGL((gn_fused_kernel<1, 0>), dim3(B * G), dim3(GNF_THREADS), 0, stream, x, dy, gamma, beta,
                       const_cast<float*>(stats), HW, C, 0.f, silu, dres, dx, 1, nullptr, GnSlabSrc{});
    AQL_CHECK_LAUNCH("aql_groupnorm_silu_bwd");
    return AQL_OK;
  }
  int threads, rps;
  const int nslab = gn_geometry(HW, C, &threads, &rps);
  // split form: per-(sample, group, piece) sums in one launch, the coalesced second pass adds the pieces itself -- two launches,
  // no finalize (with the old per-group second pass this form measured 36 vs 32 us at 64x64x320 and was not used).  AQL_GN_BWD_SPLIT=0
  // restores the three-launch slab form
  static const int bwd_split = AQL_TUNE_INT("AQL_GN_BWD_SPLIT", 1);
  static const int apply2s = AQL_TUNE_INT("AQL_GN_APPLY2", 1);
  int ns_b = (bwd_split && apply2s && (C / G) >= 8) ? gn_split(C, HW) : 0;
  // measured (tools/tune_gn.py, 64x64): B=4: 25.1 / 32.7 / 40.2 us against 29.2 / 38.8 / 47.9 at 320 / 640 / 960 channels;
  // B=8: 44.9 / 57.2 / 73.1 against 41.3 / 58.7 / 75.3 -- the split form loses only with > 768 workgroups of 20-byte segments
  if (ns_b && (long)B * G * ns_b > 768 && C < 640 && bwd_split == 1) ns_b = 0;
  if (const int nss = (apply2s && (C / G) >= 8) ? gn_rowstats_slabs(B, HW, C, nslab) : 0) {
    const int rps_s = (HW + nss - 1) / nss, ns = (HW + rps_s - 1) / rps_s;
#ifdef AQL_EXPERIMENTS
    static const int skip_stats_b = getenv("AQL_EXP_GN_SKIPSTATS") ? atoi(getenv("AQL_EXP_GN_SKIPSTATS")) : 0;
    hipStreamCaptureStatus cstb = hipStreamCaptureStatusNone;
    if (skip_stats_b >= 2) (void)hipStreamIsCapturing(stream, &cstb);
    if (!(skip_stats_b >= 2 && cstb == hipStreamCaptureStatusActive))
#endif
    hipLaunchKernelGGL(gn_rowstats_kernel<1>, dim3(ns, B), dim3(threads), threads * 16, stream, x, dy, gamma, beta, stats, HW, C, rps_s,
                       silu, scratch);
    hipLaunchKernelGGL(gn_apply2_kernel<1>, dim3(nslab, B), dim3(threads), 0, stream, x, dy, gamma, beta, stats, nullptr, scratch, ns, 1,
                       1.f / ((float)HW * (C / G)), 0.f, nullptr, HW, C, rps, silu, dres, dx);
    AQL_CHECK_LAUNCH("aql_groupnorm_silu_bwd");
    return AQL_OK;
  }
  if (const int ns = ns_b) {
    hipLaunchKernelGGL((gn_fused_kernel<1, 1>), dim3(B * G * ns), dim3(GNF_THREADS), 0, stream, x, dy, gamma, beta,
                       const_cast<float*>(stats), HW, C, 0.f, silu, nullptr, dx, ns, scratch, GnSlabSrc{});
    hipLaunchKernelGGL(gn_apply2_kernel<1>, dim3(nslab, B), dim3(threads), 0, stream, x, dy, gamma, beta, stats, nullptr, scratch, ns, 0,
                       1.f / ((float)HW * (C / G)), 0.f, nullptr, HW, C, rps, silu, dres, dx);
    AQL_CHECK_LAUNCH("aql_groupnorm_silu_bwd");
    return AQL_OK;
  }
  float* dstats = scratch + (long)B * 128 * G * 2;
  hipLaunchKernelGGL(gn_partial_kernel<1>, dim3(nslab, B), dim3(threads), threads * 64, stream, x, dy, gamma, beta, stats, HW, C,
                     rps, silu, scratch);
  hipLaunchKernelGGL(gn_finalize_kernel<1>, dim3(B), dim3(256), 0, stream, scratch, nslab,
                     1.f / ((float)HW * (C / G)), 0.f, dstats);
  static const int apply2 = AQL_TUNE_INT("AQL_GN_APPLY2", 1);   // A/B hook
  if (apply2 && (C / G) >= 8) {
    hipLaunchKernelGGL(gn_apply2_kernel<1>, dim3(nslab, B), dim3(threads), 0, stream, x, dy, gamma, beta, stats, dstats, nullptr, 0, 0,
                       0.f, 0.f, nullptr, HW, C, rps, silu, dres, dx);
    AQL_CHECK_LAUNCH("aql_groupnorm_silu_bwd");
    return AQL_OK;
  }
  const long nchunk = (long)HW * (C / 8);
  int blocks = (int)((nchunk + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(gn_apply_kernel<1>, dim3(blocks, B), dim3(256), 0, stream, x, dy, gamma, beta, stats, dstats, HW, C,
                     silu, dres, dx);
  AQL_CHECK_LAUNCH("aql_groupnorm_silu_bwd");
  return AQL_OK;
}

// GroupNorm(+SiLU) straight behind a split-K convolution / GEMM whose finalize launch was held back (aql_conv3x3_fwd_defer,
// aql_conv3x3_bwd_data_defer; round 6): the GroupNorm's first pass sums the fp32 partial slabs [splits][B*HW][C] and applies the
// GEMM's epilogue itself.  Same results as aql_splitk_finalize + aql_groupnorm_silu_fwd / _bwd, bit for bit; one launch instead of two.
// Returns AQL_NOT_FUSED (100) for maps the one-launch GroupNorm does not take (the caller then runs aql_splitk_finalize and the
// plain entry point).  Forward: xout receives the finished conv output (saved for backward, read by residual branches).
extern "C" int aql_groupnorm_silu_fwd_slabs(const float* slabs, int splits, const bf16_t* bias, const bf16_t* rowbias, long rowbias_ld,
                                            const bf16_t* residual, bf16_t* xout, int B, int HW, int C, const bf16_t* gamma,
                                            const bf16_t* beta, float eps, int silu, bf16_t* y, float* stats, hipStream_t stream) {
  AQL_CHECK_ARG(slabs && xout && gamma && beta && y && stats && splits >= 1 && splits <= 64, "aql_groupnorm_silu_fwd_slabs: null operand / bad split count");
  AQL_CHECK_ARG(C % 8 == 0 && C % G == 0 && C / 8 <= 1024, "aql_groupnorm_silu_fwd_slabs: bad C=%d", C);
  AQL_CHECK_ARG(rowbias == nullptr || rowbias_ld >= C, "aql_groupnorm_silu_fwd_slabs: row-bias leading dimension");
  if (!gn_use_fused(C, HW)) return 100;
  GnSlabSrc ss{slabs, splits, (long)B * HW * C, bias, rowbias, rowbias_ld, residual, xout};
  hipLaunchKernelGGL((gn_fused_kernel<0, 0, 1>), dim3(B * G), dim3(GNF_THREADS), 0, stream, nullptr, nullptr, gamma, beta, stats, HW,
                     C, eps, silu, nullptr, y, 1, nullptr, ss);
  AQL_CHECK_LAUNCH("aql_groupnorm_silu_fwd_slabs");
  return AQL_OK;
}

// Backward: dy = the sum of the slabs (rounded to bf16 as the finalize launch would have stored it) is consumed in place.
extern "C" int aql_groupnorm_silu_bwd_slabs(const bf16_t* x, const float* slabs, int splits, int B, int HW, int C, const bf16_t* gamma,
                                            const bf16_t* beta, int silu, const float* stats, const bf16_t* dres, bf16_t* dx,
                                            hipStream_t stream) {
  AQL_CHECK_ARG(x && slabs && gamma && beta && dx && stats && splits >= 1 && splits <= 64, "aql_groupnorm_silu_bwd_slabs: null operand / bad split count");
  AQL_CHECK_ARG(C % 8 == 0 && C % G == 0 && C / 8 <= 1024, "aql_groupnorm_silu_bwd_slabs: bad C=%d", C);
  if (!gn_use_fused(C, HW, true)) return 100;
  GnSlabSrc ss{slabs, splits, (long)B * HW * C, nullptr, nullptr, 0, nullptr, nullptr};
  hipLaunchKernelGGL((gn_fused_kernel<1, 0, 1>), dim3(B * G), dim3(GNF_THREADS), 0, stream, x, nullptr, gamma, beta,
                     const_cast<float*>(stats), HW, C, 0.f, silu, dres, dx, 1, nullptr, ss);
  AQL_CHECK_LAUNCH("aql_groupnorm_silu_bwd_slabs");
  return AQL_OK;
}

extern "C" int aql_layernorm_fwd(const bf16_t* x, long M, int C, const bf16_t* gamma, const bf16_t* beta, float eps,
                                 bf16_t* y, float* stats, hipStream_t stream) {
  AQL_CHECK_ARG(x && gamma && beta && y && stats, "aql_layernorm_fwd: null operand");
  AQL_CHECK_ARG(C % 8 == 0 && C <= 64 * 8 * LN_MAXC, "aql_layernorm_fwd: unsupported C=%d", C);
  static const int rows = AQL_TUNE_INT("AQL_LN_ROWS", 1);   // tuning hook: rows per wavefront (2: measured equal or slower, tools/time_ln.py)
  ln_launch<0>(rows >= 2 && M >= 4096, x, nullptr, gamma, beta, stats, M, C, eps, y, stream);
  AQL_CHECK_LAUNCH("aql_layernorm_fwd");
  return AQL_OK;
}

extern "C" int aql_layernorm_bwd(const bf16_t* x, const bf16_t* dy, long M, int C, const bf16_t* gamma,
                                 const float* stats, const bf16_t* dres, bf16_t* dx, hipStream_t stream) {
  AQL_CHECK_ARG(x && dy && gamma && dx && stats, "aql_layernorm_bwd: null operand");
  AQL_CHECK_ARG(C % 8 == 0 && C <= 64 * 8 * LN_MAXC, "aql_layernorm_bwd: unsupported C=%d", C);
  static const int rows = AQL_TUNE_INT("AQL_LN_ROWS", 1);
  ln_launch<1>(rows >= 2 && M >= 4096, x, dy, gamma, dres, const_cast<float*>(stats), M, C, 0.f, dx, stream);
  AQL_CHECK_LAUNCH("aql_layernorm_bwd");
  return AQL_OK;
}
