// SecretDecoder TRAINING kernels (stage 1: train/latent_wm_pretrain.py:159-225; robustness fine-tune:
// train/rob_enhance_finetune.py, decoder fwd+bwd at B=16): torchvision efficientnet_b1 in train mode, i.e. BatchNorm
// with batch statistics, plus the backward of every layer.  fp32 like the reference (the decoder is never cast),
// channels-last activations [B,H,W,C] == [M,C].  Layers are kept separate (conv | BN+SiLU | SE | ...) because training
// needs each pre-normalisation tensor for the backward; BN backward recomputes the normalised value instead of storing it.
#include "aql_common.h"

namespace {

__device__ __forceinline__ float sigmoid_(float z) { return 1.f / (1.f + __expf(-z)); }
__device__ __forceinline__ float silu_(float z) { return z * sigmoid_(z); }
__device__ __forceinline__ float dsilu_(float z) {
  const float s = sigmoid_(z);
  return s * (1.f + z * (1.f - s));
}

inline int grid_for(long n, int cap = 8192) {
  long b = (n + 255) / 256;
  if (b > cap) b = cap;
  return (int)(b < 1 ? 1 : b);
}

// ------------------------------------------------------------------------------------------------------------------
// Generic fp32 GEMM on the exact-fp32 MFMA:  C[m,n] (+)= sum_k A(m,k) * B(n,k) (+ bias[n]),
//   A(m,k) = A[m*sam + k*sak],  B(n,k) = B[n*sbn + k*sbk]   (either stride may be 1; the loader walks the unit stride)
// 64x64 tile, K tile 32, 4 wavefronts of 32x32; grid.z splits K and combines with fp32 atomics (C pre-zeroed).
// Covers the three passes of a 1x1 convolution / linear layer:
//   forward   Y[M,Co]  = X[M,Ci] . W[Co,Ci]^T     A=X (Ci,1)   B=W (Ci,1)
//   data      dX[M,Ci] = dY[M,Co] . W[Co,Ci]      A=dY (Co,1)  B=W (1,Ci)
//   weight    dW[Co,Ci] = dY^T . X   (K = M)      A=dY (1,Co)  B=X (1,Ci)
// ------------------------------------------------------------------------------------------------------------------
// TM x TN output tile: 64 x 64 (2 x 2 wavefronts of 32 x 32) or, for N <= 32, 128 x 32 (4 x 1: no wavefront multiplies padding columns)
// LA / LB: how an operand tile is fetched -- 1 = 16-byte loads along K, 2 = 16-byte loads along m / n (transposed operand), 0 = 4-byte
// loads (any strides).  Compile-time so that an instantiation carries ONE loader per operand (all four inline cost 176 VGPRs: two
// workgroups per CU for a kernel whose only latency hiding is co-resident workgroups).
template <int TM, int TN, int LA, int LB>
__global__ __launch_bounds__(256, 4) void gemm_f32_kernel(const float* __restrict__ A, long sam, long sak,
                                                       const float* __restrict__ B, long sbn, long sbk,
                                                       const float* __restrict__ bias, float* __restrict__ C, long ldc,
                                                       long M, int N, long K, int splits) {
  __shared__ float sA[TM][33], sB[TN][33];
  constexpr int WAVES_N = TN / 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long m0 = (long)blockIdx.x * TM;
  const int n0 = blockIdx.y * TN;
  const int wm = (wave / WAVES_N) * 32, wn = (wave % WAVES_N) * 32;
  const long kt = (K + 31) / 32;
  const long k_begin = (kt * blockIdx.z / splits) * 32, k_end = min(K, (kt * (blockIdx.z + 1) / splits) * 32);
  f32x16_t acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  // 16-byte loads where an operand's unit stride is K and its rows are 16-byte aligned (the 1x1 convolutions: channel counts % 4 == 0)
  for (long k0 = k_begin; k0 < k_end; k0 += 32) {
    if constexpr (LA == 1) {
#pragma unroll
      for (int it = 0; it < TM / 32; ++it) {
        const int id = tid + it * 256, r = id >> 3, c = (id & 7) * 4;
        const long m = m0 + r, k = k0 + c;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (m < M && k + 3 < k_end) v = *reinterpret_cast<const float4*>(A + m * sam + k);
        else if (m < M) {
          if (k < k_end) v.x = A[m * sam + k];
          if (k + 1 < k_end) v.y = A[m * sam + k + 1];
          if (k + 2 < k_end) v.z = A[m * sam + k + 2];
        }
        sA[r][c] = v.x, sA[r][c + 1] = v.y, sA[r][c + 2] = v.z, sA[r][c + 3] = v.w;
      }
    } else if constexpr (LA == 2) {     // unit stride along m (weight gradients: A = dY^T): 4 consecutive rows per lane
#pragma unroll
      for (int it = 0; it < TM / 32; ++it) {
        const int id = tid + it * 256, r = (id % (TM / 4)) * 4, c = id / (TM / 4);
        const long m = m0 + r, k = k0 + c;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < k_end) {
          const float* ap = A + m + k * sak;
          if (m + 3 < M) v = *reinterpret_cast<const float4*>(ap);
          else {
            if (m < M) v.x = ap[0];
            if (m + 1 < M) v.y = ap[1];
            if (m + 2 < M) v.z = ap[2];
          }
        }
        sA[r][c] = v.x, sA[r + 1][c] = v.y, sA[r + 2][c] = v.z, sA[r + 3][c] = v.w;
      }
    } else {
#pragma unroll
      for (int it = 0; it < TM / 8; ++it) {
        const int id = tid + it * 256;
        int r, c;
        if (sak == 1) r = id >> 5, c = id & 31; else c = id / TM, r = id % TM;
        const long m = m0 + r, k = k0 + c;
        sA[r][c] = (m < M && k < k_end) ? A[m * sam + k * sak] : 0.f;
      }
    }
    if constexpr (LB == 1) {
#pragma unroll
      for (int it = 0; it < TN / 32; ++it) {
        const int id = tid + it * 256, r = id >> 3, c = (id & 7) * 4;
        const long k = k0 + c;
        const bool in = n0 + r < N;
        const float* bp = B + (long)(n0 + r) * sbn + k;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (in && k + 3 < k_end) v = *reinterpret_cast<const float4*>(bp);
        else if (in) {
          if (k < k_end) v.x = bp[0];
          if (k + 1 < k_end) v.y = bp[1];
          if (k + 2 < k_end) v.z = bp[2];
        }
        sB[r][c] = v.x, sB[r][c + 1] = v.y, sB[r][c + 2] = v.z, sB[r][c + 3] = v.w;
      }
    } else if constexpr (LB == 2) {
#pragma unroll
      for (int it = 0; it < TN / 32; ++it) {
        const int id = tid + it * 256, r = (id % (TN / 4)) * 4, c = id / (TN / 4);
        const long k = k0 + c;
        const int nn = n0 + r;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < k_end) {
          const float* bp = B + nn + k * sbk;
          if (nn + 3 < N) v = *reinterpret_cast<const float4*>(bp);
          else {
            if (nn < N) v.x = bp[0];
            if (nn + 1 < N) v.y = bp[1];
            if (nn + 2 < N) v.z = bp[2];
          }
        }
        sB[r][c] = v.x, sB[r + 1][c] = v.y, sB[r + 2][c] = v.z, sB[r + 3][c] = v.w;
      }
    } else {
#pragma unroll
      for (int it = 0; it < TN / 8; ++it) {
        const int id = tid + it * 256;
        int r, c;
        if (sbk == 1) r = id >> 5, c = id & 31; else c = id / TN, r = id % TN;
        const long kb = k0 + c;
        sB[r][c] = (n0 + r < N && kb < k_end) ? B[(long)(n0 + r) * sbn + kb * sbk] : 0.f;
      }
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 32; kk += 2) {
      // D[i][j] += A[i][k] B[k][j] with i = output row m, j = output column n: a lane holds one column n = lane & 31 of 16 rows, so
      // a store instruction writes two 128-byte row segments (the first form had the lanes along m: 64 rows x 4 bytes per store)
      const float a = sA[wm + (lane & 31)][kk + (lane >> 5)];
      const float bq = sB[wn + (lane & 31)][kk + (lane >> 5)];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bq, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  const int n = n0 + wn + (lane & 31);
  if (n >= N) return;
  const float bv = (bias != nullptr && blockIdx.z == 0) ? bias[n] : 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const long m = m0 + wm + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
    if (m >= M) continue;
    const float v = acc[e] + bv;
    if (splits > 1) atomicAdd(C + m * ldc + n, v);
    else C[m * ldc + n] = v;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// BatchNorm2d, training mode, on [M,C] with optional fused SiLU.
//   stats:    per-(row-chunk, channel) partial sums  ->  finalize (double): mean, invstd, running stats
//   apply:    y = act(gamma * (x - mean) * invstd + beta)
//   backward: z recomputed; dz = dy * act'(z);  dgamma = sum dz*xhat, dbeta = sum dz,
//             dx = gamma*invstd * (dz - dbeta/M - xhat*dgamma/M)
// A workgroup covers 64 channels x 4 row lanes; deterministic (no atomics).
// ------------------------------------------------------------------------------------------------------------------
constexpr int BN_ROWS = 2048;  // rows per partial chunk

// Lanes = 16 channel quads (16-byte loads) x 4 rows, 4 wavefronts = 16 rows per pass, four passes in flight per thread (the first
// form -- one 4-byte load per lane and row, one row in flight per wavefront -- ran the 16 x 256 x 256-pixel maps of the rob-finetune
// step at 0.8 TB/s: 26 of its 92 ms).  The four row lanes are folded with two xor-shuffles, the wavefronts through LDS, in a fixed order.
template <bool BWD, bool RS = false>
__global__ __launch_bounds__(256) void bn_partial_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                         const float* __restrict__ mean, const float* __restrict__ invstd,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         int act, long M, int C, float* __restrict__ p0,
                                                         float* __restrict__ p1, const float* __restrict__ rowscale = nullptr,
                                                         long rps = 1) {
  // rowscale (BWD, round 6): dy arrives as the gradient of  rowscale[m / rps] * act(BN(x)) + residual  (stochastic depth + skip of an
  // MBConv block folded into the BatchNorm, aql_bn_train_fwd_res): the incoming gradient is scaled per sample on the fly
  __shared__ float r0[4][64], r1[4][64];
  const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
  const int q = lane & 15, rl = lane >> 4;
  const int c = blockIdx.y * 64 + q * 4;
  const long m_begin = (long)blockIdx.x * BN_ROWS, m_end = min(M, m_begin + BN_ROWS);
  float a0[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f};
  if (c < C) {
    float4 mu = make_float4(0.f, 0.f, 0.f, 0.f), is = mu, g = mu, bt = mu;
    if (BWD) {
      mu = *reinterpret_cast<const float4*>(mean + c), is = *reinterpret_cast<const float4*>(invstd + c);
      g = *reinterpret_cast<const float4*>(gamma + c), bt = *reinterpret_cast<const float4*>(beta + c);
    }
    auto add = [&](const float4& xv, const float4& dv) __attribute__((always_inline)) {
      const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ds[4] = {dv.x, dv.y, dv.z, dv.w};
      const float ms[4] = {mu.x, mu.y, mu.z, mu.w}, ss[4] = {is.x, is.y, is.z, is.w};
      const float gs[4] = {g.x, g.y, g.z, g.w}, bs[4] = {bt.x, bt.y, bt.z, bt.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (!BWD) {
          a0[j] += xs[j];
          a1[j] += xs[j] * xs[j];
        } else {
          const float xh = (xs[j] - ms[j]) * ss[j];
          float dz = ds[j];
          if (act) dz *= dsilu_(gs[j] * xh + bs[j]);
          a0[j] += dz;
          a1[j] += dz * xh;
        }
      }
    };
    long m = m_begin + part * 4 + rl;
    for (; m + 48 < m_end; m += 64) {
      float4 xv[4], dv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        xv[u] = *reinterpret_cast<const float4*>(x + (m + 16 * u) * C + c);
        dv[u] = BWD ? *reinterpret_cast<const float4*>(dy + (m + 16 * u) * C + c) : xv[u];
        if (BWD && RS) {
          const float rs = rowscale[(uint32_t)(m + 16 * u) / (uint32_t)rps];
          dv[u].x *= rs, dv[u].y *= rs, dv[u].z *= rs, dv[u].w *= rs;
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) add(xv[u], dv[u]);
    }
    for (; m < m_end; m += 16) {
      const float4 xv = *reinterpret_cast<const float4*>(x + m * C + c);
      float4 dv = BWD ? *reinterpret_cast<const float4*>(dy + m * C + c) : xv;
      if (BWD && RS) {
        const float rs = rowscale[(uint32_t)m / (uint32_t)rps];
        dv.x *= rs, dv.y *= rs, dv.z *= rs, dv.w *= rs;
      }
      add(xv, dv);
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    a0[j] += __shfl_xor(a0[j], 16);
    a0[j] += __shfl_xor(a0[j], 32);
    a1[j] += __shfl_xor(a1[j], 16);
    a1[j] += __shfl_xor(a1[j], 32);
  }
  if (rl == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) r0[part][q * 4 + j] = a0[j], r1[part][q * 4 + j] = a1[j];
  }
  __syncthreads();
  const int cl = threadIdx.x, cc = blockIdx.y * 64 + cl;
  if (cl < 64 && cc < C) {
    p0[(long)blockIdx.x * C + cc] = (r0[0][cl] + r0[1][cl]) + (r0[2][cl] + r0[3][cl]);
    p1[(long)blockIdx.x * C + cc] = (r1[0][cl] + r1[1][cl]) + (r1[2][cl] + r1[3][cl]);
  }
}

// Column sums of the two partial tables in double precision: a workgroup = 64 channels x 4 chunk lanes (chunk i goes to lane i % 4),
// the four lanes are added in a fixed order.  (One thread per channel walking all 512 chunks took 18 us per launch, 138 launches per step.)
__device__ __forceinline__ void bn_column_sums(const float* __restrict__ p0, const float* __restrict__ p1, int nchunk, int C, int c,
                                               int part, double (&red)[2][4][64], double& s, double& q) {
  double s0 = 0.0, q0 = 0.0, s1 = 0.0, q1 = 0.0;
  if (c < C) {
    int i = part;
    for (; i + 4 < nchunk; i += 8) {
      s0 += p0[(long)i * C + c], q0 += p1[(long)i * C + c];
      s1 += p0[(long)(i + 4) * C + c], q1 += p1[(long)(i + 4) * C + c];
    }
    for (; i < nchunk; i += 4) s0 += p0[(long)i * C + c], q0 += p1[(long)i * C + c];
  }
  const int cl = threadIdx.x & 63;
  red[0][part][cl] = s0 + s1;
  red[1][part][cl] = q0 + q1;
  __syncthreads();
  s = (red[0][0][cl] + red[0][1][cl]) + (red[0][2][cl] + red[0][3][cl]);
  q = (red[1][0][cl] + red[1][1][cl]) + (red[1][2][cl] + red[1][3][cl]);
}

__global__ __launch_bounds__(256) void bn_finalize_fwd_kernel(const float* __restrict__ p0, const float* __restrict__ p1,
                                                              int nchunk, long M, int C, float eps, float momentum,
                                                              float* __restrict__ mean, float* __restrict__ invstd,
                                                              float* __restrict__ run_mean, float* __restrict__ run_var) {
  __shared__ double red[2][4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
  double s, q;
  bn_column_sums(p0, p1, nchunk, C, c, part, red, s, q);
  if (part != 0 || c >= C) return;
  const double mu = s / (double)M;
  double var = q / (double)M - mu * mu;
  if (var < 0.0) var = 0.0;
  mean[c] = (float)mu;
  invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (run_mean != nullptr) {
    const double unb = M > 1 ? var * (double)M / (double)(M - 1) : var;
    run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * (float)mu;
    run_var[c] = (1.f - momentum) * run_var[c] + momentum * (float)unb;
  }
}

__global__ __launch_bounds__(256) void bn_finalize_bwd_kernel(const float* __restrict__ p0, const float* __restrict__ p1,
                                                              int nchunk, int C, float* __restrict__ dbeta,
                                                              float* __restrict__ dgamma) {
  __shared__ double red[2][4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
  double s, q;
  bn_column_sums(p0, p1, nchunk, C, c, part, red, s, q);
  if (part != 0 || c >= C) return;
  dbeta[c] = (float)s;
  dgamma[c] = (float)q;
}

template <bool BWD, bool RS = false>
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                       const float* __restrict__ mean, const float* __restrict__ invstd,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ dbeta, const float* __restrict__ dgamma,
                                                       int act, long M, int C, float* __restrict__ out,
                                                       const float* __restrict__ res = nullptr,
                                                       const float* __restrict__ rowscale = nullptr, long rps = 1) {
  // forward with res / rowscale (round 6):  out = rowscale[m / rps] * act(BN(x)) + res  -- the stochastic-depth scale and the skip
  // connection of an MBConv block in the BatchNorm's apply pass (two element-wise passes fewer); backward: dy scaled per sample
  const long n = M * (C / 4);
  const float invM = 1.f / (float)M;
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < n; id += (long)gridDim.x * blockDim.x) {
    const int c = (int)(id % (C / 4)) * 4;
    const long off = (id / (C / 4)) * C + c;
    const float4 xv = *reinterpret_cast<const float4*>(x + off);
    const float4 mu = *reinterpret_cast<const float4*>(mean + c), is = *reinterpret_cast<const float4*>(invstd + c);
    const float4 g = *reinterpret_cast<const float4*>(gamma + c), bt = *reinterpret_cast<const float4*>(beta + c);
    const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, mus[4] = {mu.x, mu.y, mu.z, mu.w}, iss[4] = {is.x, is.y, is.z, is.w};
    const float gs[4] = {g.x, g.y, g.z, g.w}, bs[4] = {bt.x, bt.y, bt.z, bt.w};
    float o[4];
    float rsc = 1.f;
    if constexpr (RS) {
      if (rowscale != nullptr) rsc = rowscale[(uint32_t)(id / (C / 4)) / (uint32_t)rps];      // (M < 2^32 rows: the entry points check)
    }
    if (!BWD) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float z = gs[i] * (xs[i] - mus[i]) * iss[i] + bs[i];
        o[i] = act ? silu_(z) : z;
      }
      if (RS && rowscale != nullptr) {
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = __fmul_rn(o[i], rsc);      // (two roundings, as the two launches it replaces)
      }
      if (RS && res != nullptr) {
        const float4 rv = *reinterpret_cast<const float4*>(res + off);
        o[0] = __fadd_rn(o[0], rv.x), o[1] = __fadd_rn(o[1], rv.y), o[2] = __fadd_rn(o[2], rv.z), o[3] = __fadd_rn(o[3], rv.w);
      }
    } else {
      float4 dv = *reinterpret_cast<const float4*>(dy + off);
      if constexpr (RS) dv.x *= rsc, dv.y *= rsc, dv.z *= rsc, dv.w *= rsc;
      const float4 db = *reinterpret_cast<const float4*>(dbeta + c), dg = *reinterpret_cast<const float4*>(dgamma + c);
      const float ds[4] = {dv.x, dv.y, dv.z, dv.w}, dbs[4] = {db.x, db.y, db.z, db.w}, dgs[4] = {dg.x, dg.y, dg.z, dg.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float xh = (xs[i] - mus[i]) * iss[i];
        float dz = ds[i];
        if (act) dz *= dsilu_(gs[i] * xh + bs[i]);
        o[i] = gs[i] * iss[i] * (dz - dbs[i] * invM - xh * dgs[i] * invM);
      }
    }
    *reinterpret_cast<float4*>(out + off) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// depthwise k x k conv (k = 3|5, stride 1|2, pad k/2), NHWC, w packed [k*k][C].  MODE 0: y = conv(x);  MODE 1: dx =
// adjoint(dy) (gather form: for every input pixel, the outputs that read it).
// ------------------------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void dw_kernel(const float* __restrict__ src, const float* __restrict__ w, int B, int H,
                                                 int W, int C, int k, int stride, float* __restrict__ dst) {
  const int pad = k / 2;
  const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
  const int c4n = C / 4;
  const int Hd = MODE == 0 ? Ho : H, Wd = MODE == 0 ? Wo : W;  // extent of dst
  const long n = (long)B * Hd * Wd * c4n;
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < n; id += (long)gridDim.x * blockDim.x) {
    const int c = (int)(id % c4n) * 4;
    long p = id / c4n;
    const int xd = (int)(p % Wd);
    p /= Wd;
    const int yd = (int)(p % Hd);
    const int b = (int)(p / Hd);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int kh = 0; kh < k; ++kh) {
      int ys;
      if (MODE == 0) {
        ys = yd * stride + kh - pad;
        if (ys < 0 || ys >= H) continue;
      } else {  // yd = yo*stride + kh - pad  ->  yo = (yd + pad - kh) / stride
        const int t = yd + pad - kh;
        if (t < 0 || t % stride != 0) continue;
        ys = t / stride;
        if (ys >= Ho) continue;
      }
      for (int kw = 0; kw < k; ++kw) {
        int xs;
        if (MODE == 0) {
          xs = xd * stride + kw - pad;
          if (xs < 0 || xs >= W) continue;
        } else {
          const int t = xd + pad - kw;
          if (t < 0 || t % stride != 0) continue;
          xs = t / stride;
          if (xs >= Wo) continue;
        }
        const int Hs = MODE == 0 ? H : Ho, Ws = MODE == 0 ? W : Wo;
        const float4 v = *reinterpret_cast<const float4*>(src + (((long)b * Hs + ys) * Ws + xs) * C + c);
        const float4 ww = *reinterpret_cast<const float4*>(w + (long)(kh * k + kw) * C + c);
        acc.x += v.x * ww.x;
        acc.y += v.y * ww.y;
        acc.z += v.z * ww.z;
        acc.w += v.w * ww.w;
      }
    }
    *reinterpret_cast<float4*>(dst + (((long)b * Hd + yd) * Wd + xd) * C + c) = acc;
  }
}

// dw[tap][c] += sum_{b,yo,xo} dy[b,yo,xo,c] * x[b, yo*s+kh-pad, xo*s+kw-pad, c].  A workgroup = 64 channels x `rows` output rows:
// lanes = 16 channel quads (16-byte loads) x 4 pixels, 4 wavefronts = 16 consecutive output pixels of one row per pass; all K*K tap
// sums of a lane's 4 channels in registers (dy read once per pixel, the K*K input reads hit L1: neighbouring pixels share them), no
// division per pixel (the row is the loop, not a flat pixel index).  The first form (4-byte loads, one pixel per wavefront and
// pass, three integer divisions per pixel) took 22 of the 92 ms of the rob-finetune step.  One atomicAdd per (workgroup, tap, channel).
template <int K>
__global__ __launch_bounds__(256) void dw_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, int B,
                                                       int H, int W, int C, int stride, int rows,
                                                       float* __restrict__ dw) {
  __shared__ float red[4][64];
  constexpr int pad = K / 2;
  const int Ho = (H + 2 * pad - K) / stride + 1, Wo = (W + 2 * pad - K) / stride + 1;
  const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
  const int q = lane & 15, pl = lane >> 4;
  const int c = blockIdx.y * 64 + q * 4;
  const long R = (long)B * Ho;
  const long r_begin = (long)blockIdx.x * rows, r_end = min(R, r_begin + rows);
  float4 acc[K * K];
#pragma unroll
  for (int t = 0; t < K * K; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < C)
    for (long r = r_begin; r < r_end; ++r) {
      const int yo = (int)(r % Ho);
      const long b = r / Ho;
      const int y0 = yo * stride - pad;
      const float* dyr = dy + r * Wo * (long)C + c;
      const float* xb = x + b * H * (long)W * C + c;
      for (int xo = part * 4 + pl; xo < Wo; xo += 16) {
        const float4 g = *reinterpret_cast<const float4*>(dyr + (long)xo * C);
        const int x0 = xo * stride - pad;
#pragma unroll
        for (int kh = 0; kh < K; ++kh) {
          const int yi = y0 + kh;
          if (yi < 0 || yi >= H) continue;
          const float* xr = xb + (long)yi * W * C;
#pragma unroll
          for (int kw = 0; kw < K; ++kw) {
            const int xi = x0 + kw;
            if (xi < 0 || xi >= W) continue;
            const float4 v = *reinterpret_cast<const float4*>(xr + (long)xi * C);
            float4& a = acc[kh * K + kw];
            a.x += g.x * v.x, a.y += g.y * v.y, a.z += g.z * v.z, a.w += g.w * v.w;
          }
        }
      }
    }
#pragma unroll
  for (int t = 0; t < K * K; ++t) {
    float v[4] = {acc[t].x, acc[t].y, acc[t].z, acc[t].w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v[j] += __shfl_xor(v[j], 16);
      v[j] += __shfl_xor(v[j], 32);
    }
    if (pl == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) red[part][q * 4 + j] = v[j];
    }
    __syncthreads();
    const int cl = threadIdx.x, cc = blockIdx.y * 64 + cl;
    if (cl < 64 && cc < C) atomicAdd(dw + (long)t * C + cc, (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]));
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------------------------
// stem 3x3 stride-2 pad-1 conv 3 -> Cout, NHWC, w packed [27][Cout] (tap-major (kh,kw,ci)).
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stem_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, int B,
                                                       int H, int W, int Cout, float* __restrict__ y) {
  __shared__ float sw[27 * 64];
  for (int i = threadIdx.x; i < 27 * Cout; i += blockDim.x) sw[i] = w[i];
  __syncthreads();
  const int Ho = H / 2, Wo = W / 2;
  const long n = (long)B * Ho * Wo * (Cout / 4);
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < n; id += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(id % (Cout / 4)) * 4;
    long p = id / (Cout / 4);
    const int xo = (int)(p % Wo);
    p /= Wo;
    const int yo = (int)(p % Ho);
    const int b = (int)(p / Ho);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int kh = 0; kh < 3; ++kh) {
      const int yi = yo * 2 + kh - 1;
      if (yi < 0 || yi >= H) continue;
      for (int kw = 0; kw < 3; ++kw) {
        const int xi = xo * 2 + kw - 1;
        if (xi < 0 || xi >= W) continue;
        const float* px = x + (((long)b * H + yi) * W + xi) * 3;
#pragma unroll
        for (int ci = 0; ci < 3; ++ci) {
          const float v = px[ci];
          const float* ww = sw + ((kh * 3 + kw) * 3 + ci) * Cout + c4;
          a0 += v * ww[0], a1 += v * ww[1], a2 += v * ww[2], a3 += v * ww[3];
        }
      }
    }
    *reinterpret_cast<float4*>(y + (((long)b * Ho + yo) * Wo + xo) * Cout + c4) = make_float4(a0, a1, a2, a3);
  }
}

// dx[b,yi,xi,ci] = sum over outputs (yo,xo) reading it, over co: dy[b,yo,xo,co] * w[(kh,kw,ci)][co]
__global__ __launch_bounds__(256) void stem_bwd_data_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                            int B, int H, int W, int Cout, float* __restrict__ dx) {
  __shared__ float sw[27 * 64];
  for (int i = threadIdx.x; i < 27 * Cout; i += blockDim.x) sw[i] = w[i];
  __syncthreads();
  const int Ho = H / 2, Wo = W / 2;
  const long n = (long)B * H * W;
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < n; id += (long)gridDim.x * blockDim.x) {
    const int xi = (int)(id % W), yi = (int)((id / W) % H);
    const long b = id / ((long)W * H);
    float a[3] = {0.f, 0.f, 0.f};
    for (int kh = 0; kh < 3; ++kh) {
      const int t = yi + 1 - kh;
      if (t < 0 || (t & 1)) continue;
      const int yo = t >> 1;
      if (yo >= Ho) continue;
      for (int kw = 0; kw < 3; ++kw) {
        const int u = xi + 1 - kw;
        if (u < 0 || (u & 1)) continue;
        const int xo = u >> 1;
        if (xo >= Wo) continue;
        const float* g = dy + ((b * Ho + yo) * Wo + xo) * Cout;
        const float* ww = sw + (kh * 3 + kw) * 3 * Cout;
        for (int co = 0; co < Cout; ++co) {
          const float gv = g[co];
          a[0] += gv * ww[co], a[1] += gv * ww[Cout + co], a[2] += gv * ww[2 * Cout + co];
        }
      }
    }
    dx[id * 3 + 0] = a[0], dx[id * 3 + 1] = a[1], dx[id * 3 + 2] = a[2];
  }
}

// dw[(kh,kw,ci)][co] += sum_p dy[p,co] * x[..]; workgroup = Cout (<=64) channels x 4 lanes over a chunk of output pixels,
// one pass with the 27 tap sums in registers
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, int B,
                                                         int H, int W, int Cout, int chunk, float* __restrict__ dw) {
  __shared__ float red[8][64];
  const int Ho = H / 2, Wo = W / 2;
  // Cout <= 32 (EfficientNet-B1: 32): 8 pixels per pass, every lane busy
  const int cw = Cout <= 32 ? 32 : 64, nparts = 256 / cw;
  const int co = threadIdx.x % cw, part = threadIdx.x / cw;
  const long P = (long)B * Ho * Wo;
  const long p_begin = (long)blockIdx.x * chunk, p_end = min(P, p_begin + chunk);
  float acc[27];
#pragma unroll
  for (int t = 0; t < 27; ++t) acc[t] = 0.f;
  if (co < Cout) {
    long p = p_begin + part;
    int xo = (int)(p % Wo), yo = (int)((p / Wo) % Ho);
    long b = p / ((long)Wo * Ho);
    for (; p < p_end; p += nparts) {
      const float g = dy[p * Cout + co];
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int yi = yo * 2 + kh - 1;
        if (yi < 0 || yi >= H) continue;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int xi = xo * 2 + kw - 1;
          if (xi < 0 || xi >= W) continue;
          const float* px = x + ((b * H + yi) * W + xi) * 3;
#pragma unroll
          for (int ci = 0; ci < 3; ++ci) acc[(kh * 3 + kw) * 3 + ci] += g * px[ci];
        }
      }
      xo += nparts;                      // next pixel of this lane without a division (nparts <= Wo)
      if (xo >= Wo) {
        xo -= Wo;
        if (++yo == Ho) yo = 0, ++b;
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 27; ++t) {
    red[part][co] = acc[t];
    __syncthreads();
    if (part == 0 && co < Cout) {
      float v = 0.f;
      for (int q = 0; q < nparts; ++q) v += red[q][co];
      atomicAdd(dw + (long)t * Cout + co, v);
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------------------------
// squeeze-excite plumbing and small elementwise pieces
// ------------------------------------------------------------------------------------------------------------------
// y[b,p,c] = x[b,p,c] * g[b,c]   (also the backward dx = dy * g)
__global__ __launch_bounds__(256) void chan_scale_kernel(const float* __restrict__ x, const float* __restrict__ g, long HW,
                                                         int C, long n4, float* __restrict__ y) {
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < n4; id += (long)gridDim.x * blockDim.x) {
    const int c = (int)(id % (C / 4)) * 4;
    const long row = id / (C / 4);
    const long b = row / HW;
    const float4 v = *reinterpret_cast<const float4*>(x + row * C + c);
    const float4 s = *reinterpret_cast<const float4*>(g + b * C + c);
    *reinterpret_cast<float4*>(y + row * C + c) = make_float4(v.x * s.x, v.y * s.y, v.z * s.z, v.w * s.w);
  }
}

// out[b,c] = scale * sum_p a[b,p,c] * (bmul ? bmul[b,p,c] : 1)   (avgpool forward: scale = 1/HW; gate gradient: a=dy, bmul=x)
// grid (C/64, B, nsplit): a workgroup reduces a slice of HW rows; nsplit > 1 combines with atomics (out pre-zeroed)
__global__ __launch_bounds__(256) void chan_reduce_kernel(const float* __restrict__ a, const float* __restrict__ bmul,
                                                          long HW, int C, float scale, int nsplit,
                                                          float* __restrict__ out) {
  __shared__ float red[4][64];
  const int b = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
  const long per = (HW + nsplit - 1) / nsplit;
  const long p0 = (long)blockIdx.z * per, p1 = min(HW, p0 + per);
  float acc = 0.f;
  if (c < C)
    for (long p = p0 + part; p < p1; p += 4) {
      const long o = ((long)b * HW + p) * C + c;
      acc += bmul ? a[o] * bmul[o] : a[o];
    }
  red[part][threadIdx.x & 63] = acc;
  __syncthreads();
  if (part == 0 && c < C) {
    const int l = threadIdx.x;
    const float v = ((red[0][l] + red[1][l]) + (red[2][l] + red[3][l])) * scale;
    if (nsplit > 1) atomicAdd(out + (long)b * C + c, v);
    else out[(long)b * C + c] = v;
  }
}

// the same reduction on 16-byte loads (C % 4 == 0): lanes = 16 channel quads x 4 rows, 16 rows per pass, four passes in flight
__global__ __launch_bounds__(256) void chan_reduce4_kernel(const float* __restrict__ a, const float* __restrict__ bmul,
                                                           long HW, int C, float scale, int nsplit,
                                                           float* __restrict__ out) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, part = threadIdx.x >> 6, q = lane & 15, rl = lane >> 4;
  const int b = blockIdx.y, c = blockIdx.x * 64 + q * 4;
  const long per = (HW + nsplit - 1) / nsplit;
  const long p0 = (long)blockIdx.z * per, p1 = min(HW, p0 + per);
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  if (c < C) {
    const float* ab = a + (long)b * HW * C + c;
    const float* mb = bmul ? bmul + (long)b * HW * C + c : nullptr;
    auto add = [&](const float4& v, const float4& w) __attribute__((always_inline)) {
      acc[0] += v.x * w.x, acc[1] += v.y * w.y, acc[2] += v.z * w.z, acc[3] += v.w * w.w;
    };
    const float4 one = make_float4(1.f, 1.f, 1.f, 1.f);
    long p = p0 + part * 4 + rl;
    for (; p + 48 < p1; p += 64) {
      float4 v[4], w[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        v[u] = *reinterpret_cast<const float4*>(ab + (p + 16 * u) * C);
        w[u] = mb ? *reinterpret_cast<const float4*>(mb + (p + 16 * u) * C) : one;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) add(v[u], w[u]);
    }
    for (; p < p1; p += 16) add(*reinterpret_cast<const float4*>(ab + p * C), mb ? *reinterpret_cast<const float4*>(mb + p * C) : one);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    acc[j] += __shfl_xor(acc[j], 16);
    acc[j] += __shfl_xor(acc[j], 32);
  }
  if (rl == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) red[part][q * 4 + j] = acc[j];
  }
  __syncthreads();
  const int l = threadIdx.x, cc = blockIdx.x * 64 + l;
  if (l < 64 && cc < C) {
    const float v = ((red[0][l] + red[1][l]) + (red[2][l] + red[3][l])) * scale;
    if (nsplit > 1) atomicAdd(out + (long)b * C + cc, v);
    else out[(long)b * C + cc] = v;
  }
}

// dx[b,p,c] (+)= g[b,c] * scale   (avgpool backward: scale = 1/HW); accumulate adds onto an existing dx
__global__ __launch_bounds__(256) void chan_bcast_kernel(const float* __restrict__ g, long HW, int C, float scale,
                                                         int accumulate, long n4, float* __restrict__ dx) {
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < n4; id += (long)gridDim.x * blockDim.x) {
    const int c = (int)(id % (C / 4)) * 4;
    const long row = id / (C / 4);
    const long b = row / HW;
    const float4 s = *reinterpret_cast<const float4*>(g + b * C + c);
    float4 v = make_float4(s.x * scale, s.y * scale, s.z * scale, s.w * scale);
    if (accumulate) {
      const float4 o = *reinterpret_cast<const float4*>(dx + row * C + c);
      v.x += o.x, v.y += o.y, v.z += o.z, v.w += o.w;
    }
    *reinterpret_cast<float4*>(dx + row * C + c) = v;
  }
}

// small-tensor activation: kind 1 = SiLU, 2 = sigmoid.  fwd: y = act(x).  bwd: y = dy * act'(x)
__global__ __launch_bounds__(256) void act_kernel(const float* __restrict__ x, const float* __restrict__ dy, int kind,
                                                  long n, float* __restrict__ y) {
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < n; id += (long)gridDim.x * blockDim.x) {
    const float z = x[id];
    if (dy == nullptr) y[id] = kind == 1 ? silu_(z) : sigmoid_(z);
    else {
      const float s = sigmoid_(z);
      y[id] = dy[id] * (kind == 1 ? s * (1.f + z * (1.f - s)) : s * (1.f - s));
    }
  }
}

// adjoint of resize_bilinear NCHW -> NHWC: dx[b,c,y,x] += w * dy[b,yo,xo,c]   (dx pre-zeroed)
__global__ __launch_bounds__(256) void resize_bwd_kernel(const float* __restrict__ dy, int B, int C, int H, int W, int Ho,
                                                         int Wo, float* __restrict__ dx) {
  const long n = (long)B * Ho * Wo;
  const float sh = (float)H / Ho, sw = (float)W / Wo;
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < n; id += (long)gridDim.x * blockDim.x) {
    const int xo = (int)(id % Wo), yo = (int)((id / Wo) % Ho);
    const int b = (int)(id / ((long)Wo * Ho));
    const float fy = fmaxf((yo + 0.5f) * sh - 0.5f, 0.f), fx = fmaxf((xo + 0.5f) * sw - 0.5f, 0.f);
    const int y0 = min((int)fy, H - 1), x0 = min((int)fx, W - 1);
    const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
    const float wy = fy - y0, wx = fx - x0;
    for (int c = 0; c < C; ++c) {
      const float g = dy[id * C + c];
      float* p = dx + ((long)b * C + c) * H * W;
      atomicAdd(p + (long)y0 * W + x0, g * (1.f - wy) * (1.f - wx));
      atomicAdd(p + (long)y0 * W + x1, g * (1.f - wy) * wx);
      atomicAdd(p + (long)y1 * W + x0, g * wy * (1.f - wx));
      atomicAdd(p + (long)y1 * W + x1, g * wy * wx);
    }
  }
}

// F.binary_cross_entropy_with_logits(logits, target) (mean) and its gradient (latent_wm_pretrain.py:196)
__global__ __launch_bounds__(256) void bce_kernel(const float* __restrict__ z, const float* __restrict__ t, long n,
                                                  float* __restrict__ loss, float* __restrict__ dz) {
  __shared__ float red[4];
  float acc = 0.f;
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < n; id += (long)gridDim.x * blockDim.x) {
    const float x = z[id], y = t[id];
    acc += fmaxf(x, 0.f) - x * y + log1pf(__expf(-fabsf(x)));
    if (dz != nullptr) dz[id] = (sigmoid_(x) - y) / (float)n;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(loss, ((red[0] + red[1]) + (red[2] + red[3])) / (float)n);
}

}  // namespace

extern "C" int aql_gemm_f32(const float* A, long sam, long sak, const float* B, long sbn, long sbk, const float* bias,
                            float* C, long ldc, long M, int N, long K, hipStream_t stream) {
  AQL_CHECK_ARG(A && B && C && M > 0 && N > 0 && K > 0 && (sam == 1 || sak == 1) && (sbn == 1 || sbk == 1),
                "aql_gemm_f32: bad args (one stride of each operand must be 1)");
  const int tn = (N <= 32 && M >= 128) ? 32 : 64;      // narrow outputs: 128 x 32 tiles
  const long tiles = tn == 32 ? (M + 127) / 128 : ((M + 63) / 64) * ((N + 63) / 64);
  int splits = 1;
  // Few output tiles under a long contraction (weight gradients: K = pixels; squeeze-excite linears: one tile): cut K so that the
  // chip holds ~8 workgroups per CU -- a workgroup has ONE K tile in flight (load, barrier, 16 MFMAs, barrier), the overlap comes
  // from co-resident workgroups.  (The first rule, <= 256 splits from K = 2048 on, ran the 1M-pixel weight gradients at 0.26 TB/s.)
  if (tiles < 512 && K >= 512) {
    splits = (int)((2048 + tiles - 1) / tiles);
    const long kt = (K + 31) / 32;
    if (splits > kt / 4) splits = (int)(kt / 4);
    if (splits > 4096) splits = 4096;
    if (splits < 1) splits = 1;
  }
  if (splits > 1) {
    if (ldc == N) (void)hipMemsetAsync(C, 0, (size_t)M * N * sizeof(float), stream);
    else (void)hipMemset2DAsync(C, ldc * sizeof(float), 0, N * sizeof(float), M, stream);
  }
  // 16-byte loads where an operand's unit stride and its row pitch allow them (the 1x1 convolutions: channel counts % 4 == 0)
  const bool al_a = (reinterpret_cast<uintptr_t>(A) & 15) == 0, al_b = (reinterpret_cast<uintptr_t>(B) & 15) == 0;
  const int la = (sak == 1 && (sam & 3) == 0 && al_a) ? 1 : (sam == 1 && sak != 1 && (sak & 3) == 0 && al_a) ? 2 : 0;
  const int lb = (sbk == 1 && (sbn & 3) == 0 && al_b) ? 1 : (sbn == 1 && sbk != 1 && (sbk & 3) == 0 && al_b) ? 2 : 0;
  const dim3 grid = tn == 32 ? dim3((unsigned)((M + 127) / 128), 1, splits) : dim3((unsigned)((M + 63) / 64), (N + 63) / 64, splits);
#define AQL_GEMM_F32_GO(TMv, TNv, LAv, LBv) \
  hipLaunchKernelGGL((gemm_f32_kernel<TMv, TNv, LAv, LBv>), grid, dim3(256), 0, stream, A, sam, sak, B, sbn, sbk, bias, C, ldc, M, N, K, splits)
#define AQL_GEMM_F32_LB(TMv, TNv, LAv) \
  do { if (lb == 1) AQL_GEMM_F32_GO(TMv, TNv, LAv, 1); else if (lb == 2) AQL_GEMM_F32_GO(TMv, TNv, LAv, 2); else AQL_GEMM_F32_GO(TMv, TNv, LAv, 0); } while (0)
#define AQL_GEMM_F32_LA(TMv, TNv) \
  do { if (la == 1) AQL_GEMM_F32_LB(TMv, TNv, 1); else if (la == 2) AQL_GEMM_F32_LB(TMv, TNv, 2); else AQL_GEMM_F32_LB(TMv, TNv, 0); } while (0)
  if (tn == 32) AQL_GEMM_F32_LA(128, 32);
  else AQL_GEMM_F32_LA(64, 64);
#undef AQL_GEMM_F32_LA
#undef AQL_GEMM_F32_LB
#undef AQL_GEMM_F32_GO
  AQL_CHECK_LAUNCH("aql_gemm_f32");
  return AQL_OK;
}

extern "C" long aql_bn_scratch_floats(long M, int C) { return 2 * ((M + BN_ROWS - 1) / BN_ROWS) * (long)C; }

// y = act(BN_train(x)); writes mean/invstd [C] for the backward and updates the running statistics (may be null)
extern "C" int aql_bn_train_fwd(const float* x, const float* gamma, const float* beta, long M, int C, float eps,
                                float momentum, int act, float* y, float* mean, float* invstd, float* run_mean,
                                float* run_var, float* scratch, hipStream_t stream) {
  AQL_CHECK_ARG(x && gamma && beta && y && mean && invstd && scratch && M > 0 && C % 4 == 0, "aql_bn_train_fwd: bad args");
  const int nchunk = (int)((M + BN_ROWS - 1) / BN_ROWS);
  float* p0 = scratch;
  float* p1 = scratch + (long)nchunk * C;
  hipLaunchKernelGGL(bn_partial_kernel<false>, dim3(nchunk, (C + 63) / 64), dim3(256), 0, stream, x, nullptr, nullptr,
                     nullptr, nullptr, nullptr, 0, M, C, p0, p1);
  hipLaunchKernelGGL(bn_finalize_fwd_kernel, dim3((C + 63) / 64), dim3(256), 0, stream, p0, p1, nchunk, M, C, eps,
                     momentum, mean, invstd, run_mean, run_var);
  hipLaunchKernelGGL(bn_apply_kernel<false>, dim3(grid_for(M * (C / 4))), dim3(256), 0, stream, x, nullptr, mean, invstd,
                     gamma, beta, nullptr, nullptr, act, M, C, y);
  AQL_CHECK_LAUNCH("aql_bn_train_fwd");
  return AQL_OK;
}

extern "C" int aql_bn_train_bwd(const float* x, const float* dy, const float* gamma, const float* beta, const float* mean,
                                const float* invstd, long M, int C, int act, float* dx, float* dgamma, float* dbeta,
                                float* scratch, hipStream_t stream) {
  AQL_CHECK_ARG(x && dy && gamma && beta && mean && invstd && dx && dgamma && dbeta && scratch && C % 4 == 0,
                "aql_bn_train_bwd: bad args");
  const int nchunk = (int)((M + BN_ROWS - 1) / BN_ROWS);
  float* p0 = scratch;
  float* p1 = scratch + (long)nchunk * C;
  hipLaunchKernelGGL(bn_partial_kernel<true>, dim3(nchunk, (C + 63) / 64), dim3(256), 0, stream, x, dy, mean, invstd,
                     gamma, beta, act, M, C, p0, p1);
  hipLaunchKernelGGL(bn_finalize_bwd_kernel, dim3((C + 63) / 64), dim3(256), 0, stream, p0, p1, nchunk, C, dbeta,
                     dgamma);
  hipLaunchKernelGGL(bn_apply_kernel<true>, dim3(grid_for(M * (C / 4))), dim3(256), 0, stream, x, dy, mean, invstd, gamma,
                     beta, dbeta, dgamma, act, M, C, dx);
  AQL_CHECK_LAUNCH("aql_bn_train_bwd");
  return AQL_OK;
}

// Round 6 (the first fusion of the decoder step).  The last BatchNorm of an MBConv block with a skip connection
// (torchvision MBConv.forward: result = stochastic_depth(block(x)); result += x; utils/models.py:84-96 builds efficientnet_b1):
//   y = rowscale[m / rows_per_sample] * BN_train(x) + res        (rowscale: the per-sample survival factor, "row" mode; res: the skip)
// in the BatchNorm's apply pass -- the chan-scale launch and the residual-add launch (5 map accesses) become 1 extra read -- and its
// backward: the gradient of x given dy = d(y) (d(res) = dy is the caller's), with the per-sample scale applied on the fly.
extern "C" int aql_bn_train_fwd_res(const float* x, const float* gamma, const float* beta, long M, int C, float eps, float momentum,
                                    int act, const float* res, const float* rowscale, long rows_per_sample, float* y, float* mean,
                                    float* invstd, float* run_mean, float* run_var, float* scratch, hipStream_t stream) {
  AQL_CHECK_ARG(x && gamma && beta && y && mean && invstd && scratch && M > 0 && C % 4 == 0 && rows_per_sample > 0 &&
                    M % rows_per_sample == 0 && M < (1L << 32), "aql_bn_train_fwd_res: bad args");
  const int nchunk = (int)((M + BN_ROWS - 1) / BN_ROWS);
  float* p0 = scratch;
  float* p1 = scratch + (long)nchunk * C;
  hipLaunchKernelGGL(bn_partial_kernel<false>, dim3(nchunk, (C + 63) / 64), dim3(256), 0, stream, x, nullptr, nullptr,
                     nullptr, nullptr, nullptr, 0, M, C, p0, p1, nullptr, 1L);
  hipLaunchKernelGGL(bn_finalize_fwd_kernel, dim3((C + 63) / 64), dim3(256), 0, stream, p0, p1, nchunk, M, C, eps,
                     momentum, mean, invstd, run_mean, run_var);
  hipLaunchKernelGGL((bn_apply_kernel<false, true>), dim3(grid_for(M * (C / 4))), dim3(256), 0, stream, x, nullptr, mean, invstd,
                     gamma, beta, nullptr, nullptr, act, M, C, y, res, rowscale, rows_per_sample);
  AQL_CHECK_LAUNCH("aql_bn_train_fwd_res");
  return AQL_OK;
}

extern "C" int aql_bn_train_bwd_rs(const float* x, const float* dy, const float* gamma, const float* beta, const float* mean,
                                   const float* invstd, long M, int C, int act, const float* rowscale, long rows_per_sample,
                                   float* dx, float* dgamma, float* dbeta, float* scratch, hipStream_t stream) {
  AQL_CHECK_ARG(x && dy && gamma && beta && mean && invstd && dx && dgamma && dbeta && scratch && C % 4 == 0 && rows_per_sample > 0 && M < (1L << 32),
                "aql_bn_train_bwd_rs: bad args");
  const int nchunk = (int)((M + BN_ROWS - 1) / BN_ROWS);
  float* p0 = scratch;
  float* p1 = scratch + (long)nchunk * C;
  hipLaunchKernelGGL((bn_partial_kernel<true, true>), dim3(nchunk, (C + 63) / 64), dim3(256), 0, stream, x, dy, mean, invstd,
                     gamma, beta, act, M, C, p0, p1, rowscale, rows_per_sample);
  hipLaunchKernelGGL(bn_finalize_bwd_kernel, dim3((C + 63) / 64), dim3(256), 0, stream, p0, p1, nchunk, C, dbeta,
                     dgamma);
  hipLaunchKernelGGL((bn_apply_kernel<true, true>), dim3(grid_for(M * (C / 4))), dim3(256), 0, stream, x, dy, mean, invstd, gamma,
                     beta, dbeta, dgamma, act, M, C, dx, nullptr, rowscale, rows_per_sample);
  AQL_CHECK_LAUNCH("aql_bn_train_bwd_rs");
  return AQL_OK;
}

// mode 0: y = dwconv(x) [B,Ho,Wo,C];  mode 1: src = dy [B,Ho,Wo,C] -> dst = dx [B,H,W,C];  mode 2: dst = dw [k*k][C]
// (+= : zeroed here) from src = x and src2 = dy
extern "C" int aql_dwconv_train(const float* src, const float* src2, const float* w, int B, int H, int W, int C, int k,
                                int stride, int mode, float* dst, hipStream_t stream) {
  AQL_CHECK_ARG(src && dst && C % 4 == 0 && (k == 3 || k == 5) && (stride == 1 || stride == 2) && mode >= 0 && mode <= 2 &&
                    (mode == 2 ? src2 != nullptr : w != nullptr),
                "aql_dwconv_train: bad args");
  const int Ho = (H + 2 * (k / 2) - k) / stride + 1, Wo = (W + 2 * (k / 2) - k) / stride + 1;
  if (mode == 0) {
    hipLaunchKernelGGL(dw_kernel<0>, dim3(grid_for((long)B * Ho * Wo * (C / 4))), dim3(256), 0, stream, src, w, B, H, W, C,
                       k, stride, dst);
  } else if (mode == 1) {
    hipLaunchKernelGGL(dw_kernel<1>, dim3(grid_for((long)B * H * W * (C / 4))), dim3(256), 0, stream, src, w, B, H, W, C, k,
                       stride, dst);
  } else {
    (void)hipMemsetAsync(dst, 0, (size_t)k * k * C * sizeof(float), stream);
    const long R = (long)B * Ho;
    int rows = 1024 / Wo;                        // ~1024 output pixels per workgroup, whole rows
    if (rows < 1) rows = 1;
    while (rows > 1 && (R + rows - 1) / rows * ((C + 63) / 64) < 1024) rows >>= 1;   // small maps: enough workgroups for the chip
    const dim3 grid((unsigned)((R + rows - 1) / rows), (C + 63) / 64);
    if (k == 3) hipLaunchKernelGGL(dw_wgrad_kernel<3>, grid, dim3(256), 0, stream, src, src2, B, H, W, C, stride, rows, dst);
    else hipLaunchKernelGGL(dw_wgrad_kernel<5>, grid, dim3(256), 0, stream, src, src2, B, H, W, C, stride, rows, dst);
  }
  AQL_CHECK_LAUNCH("aql_dwconv_train");
  return AQL_OK;
}

// mode 0: y = stem(x) ([B,H,W,3] -> [B,H/2,W/2,Cout]);  mode 1: src = dy -> dst = dx [B,H,W,3];  mode 2: dst = dw [27][Cout]
extern "C" int aql_stem_train(const float* src, const float* src2, const float* w, int B, int H, int W, int Cout, int mode,
                              float* dst, hipStream_t stream) {
  AQL_CHECK_ARG(src && dst && Cout % 4 == 0 && Cout <= 64 && H % 2 == 0 && W % 2 == 0 && mode >= 0 && mode <= 2 &&
                    (mode == 2 ? src2 != nullptr : w != nullptr),
                "aql_stem_train: bad args");
  const long P = (long)B * (H / 2) * (W / 2);
  if (mode == 0) {
    hipLaunchKernelGGL(stem_fwd_kernel, dim3(grid_for(P * (Cout / 4))), dim3(256), 0, stream, src, w, B, H, W, Cout, dst);
  } else if (mode == 1) {
    hipLaunchKernelGGL(stem_bwd_data_kernel, dim3(grid_for((long)B * H * W)), dim3(256), 0, stream, src, w, B, H, W, Cout,
                       dst);
  } else {
    (void)hipMemsetAsync(dst, 0, (size_t)27 * Cout * sizeof(float), stream);
    const int chunk = 256;     // 4096 workgroups at 16 x 256 x 256 output pixels (512 of 2048 pixels each ran 1.2 ms: two per CU)
    hipLaunchKernelGGL(stem_wgrad_kernel, dim3((unsigned)((P + chunk - 1) / chunk)), dim3(256), 0, stream, src, src2, B, H,
                       W, Cout, chunk, dst);
  }
  AQL_CHECK_LAUNCH("aql_stem_train");
  return AQL_OK;
}

// y[b,p,c] = x[b,p,c] * g[b,c]
extern "C" int aql_chan_scale(const float* x, const float* g, int B, long HW, int C, float* y, hipStream_t stream) {
  AQL_CHECK_ARG(x && g && y && C % 4 == 0, "aql_chan_scale: bad args");
  const long n4 = (long)B * HW * (C / 4);
  hipLaunchKernelGGL(chan_scale_kernel, dim3(grid_for(n4)), dim3(256), 0, stream, x, g, HW, C, n4, y);
  AQL_CHECK_LAUNCH("aql_chan_scale");
  return AQL_OK;
}

// out[b,c] = scale * sum_p a[b,p,c] * (bmul ? bmul[b,p,c] : 1)
extern "C" int aql_chan_reduce(const float* a, const float* bmul, int B, long HW, int C, float scale, float* out,
                               hipStream_t stream) {
  AQL_CHECK_ARG(a && out, "aql_chan_reduce: bad args");
  int nsplit = 1;
  const long blocks = (long)((C + 63) / 64) * B;
  if (blocks < 1024 && HW >= 512) {     // ~2048 workgroups, at least 128 rows each
    nsplit = (int)min((long)(2048 / blocks > 0 ? 2048 / blocks : 1), HW / 128);
    if (nsplit < 1) nsplit = 1;
  }
  if (nsplit > 1) (void)hipMemsetAsync(out, 0, (size_t)B * C * sizeof(float), stream);
  const bool vec = C % 4 == 0 && (reinterpret_cast<uintptr_t>(a) & 15) == 0 && (bmul == nullptr || (reinterpret_cast<uintptr_t>(bmul) & 15) == 0);
  if (vec) hipLaunchKernelGGL(chan_reduce4_kernel, dim3((C + 63) / 64, B, nsplit), dim3(256), 0, stream, a, bmul, HW, C, scale, nsplit, out);
  else hipLaunchKernelGGL(chan_reduce_kernel, dim3((C + 63) / 64, B, nsplit), dim3(256), 0, stream, a, bmul, HW, C, scale, nsplit,
                     out);
  AQL_CHECK_LAUNCH("aql_chan_reduce");
  return AQL_OK;
}

// dx[b,p,c] (+)= g[b,c] * scale
extern "C" int aql_chan_bcast(const float* g, int B, long HW, int C, float scale, int accumulate, float* dx,
                              hipStream_t stream) {
  AQL_CHECK_ARG(g && dx && C % 4 == 0, "aql_chan_bcast: bad args");
  const long n4 = (long)B * HW * (C / 4);
  hipLaunchKernelGGL(chan_bcast_kernel, dim3(grid_for(n4)), dim3(256), 0, stream, g, HW, C, scale, accumulate, n4, dx);
  AQL_CHECK_LAUNCH("aql_chan_bcast");
  return AQL_OK;
}

// kind 1 = SiLU, 2 = sigmoid; dy == null: y = act(x); else y = dy * act'(x)
extern "C" int aql_act_f32(const float* x, const float* dy, int kind, long n, float* y, hipStream_t stream) {
  AQL_CHECK_ARG(x && y && (kind == 1 || kind == 2), "aql_act_f32: bad args");
  hipLaunchKernelGGL(act_kernel, dim3(grid_for(n)), dim3(256), 0, stream, x, dy, kind, n, y);
  AQL_CHECK_LAUNCH("aql_act_f32");
  return AQL_OK;
}

// dx [B,C,H,W] = adjoint of the bilinear resize to [B,Ho,Wo,C] applied to dy
extern "C" int aql_resize_bilinear_nhwc_bwd(const float* dy, int B, int C, int H, int W, int Ho, int Wo, float* dx,
                                            hipStream_t stream) {
  AQL_CHECK_ARG(dy && dx && B > 0 && C > 0, "aql_resize_bilinear_nhwc_bwd: bad args");
  (void)hipMemsetAsync(dx, 0, (size_t)B * C * H * W * sizeof(float), stream);
  hipLaunchKernelGGL(resize_bwd_kernel, dim3(grid_for((long)B * Ho * Wo)), dim3(256), 0, stream, dy, B, C, H, W, Ho, Wo, dx);
  AQL_CHECK_LAUNCH("aql_resize_bilinear_nhwc_bwd");
  return AQL_OK;
}

// loss (device scalar, overwritten) = mean BCE-with-logits; dlogits (may be null) = d loss / d logits
extern "C" int aql_bce_logits(const float* logits, const float* target, long n, float* loss, float* dlogits,
                              hipStream_t stream) {
  AQL_CHECK_ARG(logits && target && loss && n > 0, "aql_bce_logits: bad args");
  (void)hipMemsetAsync(loss, 0, sizeof(float), stream);
  hipLaunchKernelGGL(bce_kernel, dim3(grid_for(n, 64)), dim3(256), 0, stream, logits, target, n, loss, dlogits);
  AQL_CHECK_LAUNCH("aql_bce_logits");
  return AQL_OK;
}
