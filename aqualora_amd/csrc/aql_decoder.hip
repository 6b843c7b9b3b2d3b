// SecretDecoder (reference utils/models.py:84-96 = torchvision efficientnet_b1 + Linear(1280, 2*bits)) inference kernels,
// fp32 like the reference (the decoder is never cast, train/ppft_train.py:579, 1177), channels-last activations.
// BatchNorm is folded into the preceding convolution by the host (eval mode).  The network is tiny (6.4 GFLOP/img) and
// dominated by HBM-bound depthwise / squeeze-excite / elementwise work; the 1x1 convolutions use the exact-fp32 MFMA
// (v_mfma_f32_32x32x2_f32) so that logits -- and therefore the extracted bits -- do not depend on a bf16 rounding.
#include "aql_common.h"

namespace {

__device__ __forceinline__ float silu_(float z) { return z / (1.f + __expf(-z)); }

// bilinear resize (F.interpolate(mode="bilinear", align_corners=False), models.py:92-94) NCHW -> NHWC fp32
__global__ __launch_bounds__(256) void resize_bilinear_kernel(const float* __restrict__ x, int B, int C, int H, int W,
                                                              int Ho, int Wo, float* __restrict__ y) {
  const long n = (long)B * Ho * Wo;
  const float sh = (float)H / Ho, sw = (float)W / Wo;
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < n; id += (long)gridDim.x * blockDim.x) {
    const int xo = (int)(id % Wo);
    const int yo = (int)((id / Wo) % Ho);
    const int b = (int)(id / ((long)Wo * Ho));
    float fy = fmaxf((yo + 0.5f) * sh - 0.5f, 0.f), fx = fmaxf((xo + 0.5f) * sw - 0.5f, 0.f);
    const int y0 = min((int)fy, H - 1), x0 = min((int)fx, W - 1);
    const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
    const float wy = fy - y0, wx = fx - x0;
    for (int c = 0; c < C; ++c) {
      const float* p = x + ((long)b * C + c) * H * W;
      const float v = (1.f - wy) * ((1.f - wx) * p[(long)y0 * W + x0] + wx * p[(long)y0 * W + x1]) +
                      wy * ((1.f - wx) * p[(long)y1 * W + x0] + wx * p[(long)y1 * W + x1]);
      y[id * C + c] = v;
    }
  }
}

// stem: 3x3 stride-2 pad-1 conv 3 -> Cout (<= 64) + bias + SiLU; NHWC in/out; w packed [27][Cout]
__global__ __launch_bounds__(256) void stem_conv_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, int B, int H, int W, int Cout,
                                                        float* __restrict__ y) {
  __shared__ float sw[27 * 64];
  __shared__ float sb[64];
  for (int i = threadIdx.x; i < 27 * Cout; i += blockDim.x) sw[i] = w[i];
  for (int i = threadIdx.x; i < Cout; i += blockDim.x) sb[i] = bias[i];
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
    float a0 = sb[c4], a1 = sb[c4 + 1], a2 = sb[c4 + 2], a3 = sb[c4 + 3];
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
          a0 += v * ww[0];
          a1 += v * ww[1];
          a2 += v * ww[2];
          a3 += v * ww[3];
        }
      }
    }
    *reinterpret_cast<float4*>(y + (((long)b * Ho + yo) * Wo + xo) * Cout + c4) =
        make_float4(silu_(a0), silu_(a1), silu_(a2), silu_(a3));
  }
}

// depthwise k x k (k = 3 | 5), stride 1|2, pad (k-1)/2, + bias + SiLU; NHWC, w packed [k*k][C]; C % 4 == 0
__global__ __launch_bounds__(256) void dwconv_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ bias, int B, int H, int W, int C, int k,
                                                     int stride, float* __restrict__ y) {
  const int Ho = (H + 2 * (k / 2) - k) / stride + 1, Wo = (W + 2 * (k / 2) - k) / stride + 1;
  const int c4n = C / 4;
  const long n = (long)B * Ho * Wo * c4n;
  const int pad = k / 2;
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < n; id += (long)gridDim.x * blockDim.x) {
    const int c = (int)(id % c4n) * 4;
    long p = id / c4n;
    const int xo = (int)(p % Wo);
    p /= Wo;
    const int yo = (int)(p % Ho);
    const int b = (int)(p / Ho);
    float4 acc = *reinterpret_cast<const float4*>(bias + c);
    for (int kh = 0; kh < k; ++kh) {
      const int yi = yo * stride + kh - pad;
      if (yi < 0 || yi >= H) continue;
      for (int kw = 0; kw < k; ++kw) {
        const int xi = xo * stride + kw - pad;
        if (xi < 0 || xi >= W) continue;
        const float4 v = *reinterpret_cast<const float4*>(x + (((long)b * H + yi) * W + xi) * C + c);
        const float4 ww = *reinterpret_cast<const float4*>(w + (long)(kh * k + kw) * C + c);
        acc.x += v.x * ww.x;
        acc.y += v.y * ww.y;
        acc.z += v.z * ww.z;
        acc.w += v.w * ww.w;
      }
    }
    *reinterpret_cast<float4*>(y + (((long)b * Ho + yo) * Wo + xo) * C + c) =
        make_float4(silu_(acc.x), silu_(acc.y), silu_(acc.z), silu_(acc.w));
  }
}

// global average pool NHWC [B,HW,C] -> [B,C]; one workgroup per (sample, 64-channel slab); deterministic
__global__ __launch_bounds__(256) void avgpool_kernel(const float* __restrict__ x, int HW, int C,
                                                      float* __restrict__ out) {
  __shared__ float red[4][64];
  const int b = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
  float acc = 0.f;
  if (c < C)
    for (int p = part; p < HW; p += 4) acc += x[((long)b * HW + p) * C + c];
  red[part][threadIdx.x & 63] = acc;
  __syncthreads();
  if (part == 0 && c < C) {
    const int l = threadIdx.x;
    out[(long)b * C + c] = ((red[0][l] + red[1][l]) + (red[2][l] + red[3][l])) / (float)HW;
  }
}

// the same pool cut into S pixel slabs (grid.z): raw partial sums part[b][z][c], added in slab order by the consumer (se_fc_kernel).
// One workgroup per (sample, 64-channel slab) reads 8 MB alone at the 256 x 256 level of a single image (0.5 ms per block, 12 of the
// 16.5 ms of a batch-1 extraction); with the slabs the pool is a chip-wide pass.  Deterministic: fixed slab bounds, fixed order.
__global__ __launch_bounds__(256) void avgpool_slabs_kernel(const float* __restrict__ x, int HW, int C, int S,
                                                            float* __restrict__ part) {
  __shared__ float red[4][64];
  const int b = blockIdx.y, z = blockIdx.z, c = blockIdx.x * 64 + (threadIdx.x & 63), sub = threadIdx.x >> 6;
  const int per = (HW + S - 1) / S, p0 = z * per, p1 = min(HW, p0 + per);
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  if (c < C) {
    int p = p0 + sub;
    for (; p + 12 < p1; p += 16) {      // four independent loads in flight per thread
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[u] += x[((long)b * HW + p + 4 * u) * C + c];
    }
    for (; p < p1; p += 4) acc[0] += x[((long)b * HW + p) * C + c];
  }
  red[sub][threadIdx.x & 63] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  __syncthreads();
  if (sub == 0 && c < C) {
    const int l = threadIdx.x;
    part[((long)b * S + z) * C + c] = (red[0][l] + red[1][l]) + (red[2][l] + red[3][l]);
  }
}

// squeeze-excite gates: s = sigmoid(W2 . silu(W1 . pool + b1) + b2); one workgroup per sample.  S > 0: `pool` holds S slab sums per
// channel (avgpool_slabs_kernel), the mean is formed here first.
__global__ __launch_bounds__(1024) void se_fc_kernel(const float* __restrict__ pool, int S, float inv_hw, const float* __restrict__ w1,
                                                     const float* __restrict__ b1, const float* __restrict__ w2,
                                                     const float* __restrict__ b2, int C, int Cs,
                                                     float* __restrict__ gate) {
  // 16 wavefronts; a dot product is one wavefront's job (lanes along the contraction: coalesced weight rows, xor-shuffle sum) -- a
  // thread per output walking its whole row alone took 68 us per block at C = 1920 (1.5 ms of a single image's 5 ms extraction)
  __shared__ float hid[512];
  __shared__ float mean[2048];
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  // the slab sums: many slabs go with few channels (the large maps), so the slabs are dealt to ZG thread groups per channel first
  __shared__ float zsum[1024];
  const int ZG = (S > 1 && C <= 512) ? min(S, (int)blockDim.x / C) : 1;
  if (ZG > 1) {
    const int t = threadIdx.x;
    if (t < ZG * C) {
      const int zg = t / C, c = t - zg * C;
      float a = 0.f;
      for (int z = zg; z < S; z += ZG) a += pool[((long)b * S + z) * C + c];
      zsum[t] = a;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      float a = 0.f;
      for (int zg = 0; zg < ZG; ++zg) a += zsum[zg * C + c];
      mean[c] = a * inv_hw;
    }
  } else {
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      float a = 0.f;
      if (S > 0) {
        for (int z = 0; z < S; ++z) a += pool[((long)b * S + z) * C + c];
        a *= inv_hw;
      } else {
        a = pool[(long)b * C + c];
      }
      mean[c] = a;
    }
  }
  __syncthreads();
  for (int j = wave; j < Cs; j += nw) {
    float a = 0.f;
    for (int c = lane; c < C; c += 64) a += w1[(long)j * C + c] * mean[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    if (lane == 0) hid[j] = silu_(a + b1[j]);
  }
  __syncthreads();
  // the excite linear has <= 80 inputs per output: a thread per output (a wavefront per output spends its time in the shuffle chain)
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float a0 = b2[c], a1 = 0.f;
    const float* wr = w2 + (long)c * Cs;
    int j = 0;
    for (; j + 1 < Cs; j += 2) a0 += wr[j] * hid[j], a1 += wr[j + 1] * hid[j + 1];
    if (j < Cs) a0 += wr[j] * hid[j];
    gate[(long)b * C + c] = 1.f / (1.f + __expf(-(a0 + a1)));
  }
}

// pointwise (1x1) convolution / linear, fp32 in and out, exact-fp32 MFMA:
//   y[m, n] = act( sum_k (x[m,k] * gate[m / rows_per_sample, k]) * w[n,k] + bias[n] ) + residual[m,n]
// 64x64 output tile per workgroup (4 wavefronts of 32x32), K tile 32.
__global__ __launch_bounds__(256) void pwconv_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ bias, const float* __restrict__ gate,
                                                     int rows_per_sample, const float* __restrict__ residual, long M,
                                                     int N, int K, int act, float* __restrict__ y) {
  __shared__ float sA[64][33], sB[64][33];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long m0 = (long)blockIdx.x * 64;
  const int n0 = blockIdx.y * 64;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
  f32x16_t acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  // a thread fetches the same two rows of each operand for every K tile: the sample of its rows (a 64-bit division) is found once,
  // not per element and K tile (4.7 us per K tile at 16 x 16: 280 us for the 60 tiles of the 1920 -> 320 projection of one image)
  const bool vec = (K & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(gate)) & 15) == 0;
  const float* xr[2];
  const float* gr[2];
  const float* wr[2];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int r = (tid + it * 256) >> 3;
    const long m = m0 + r;
    xr[it] = m < M ? x + m * K : nullptr;
    gr[it] = (m < M && gate != nullptr) ? gate + (m / rows_per_sample) * K : nullptr;
    wr[it] = n0 + r < N ? w + (long)(n0 + r) * K : nullptr;
  }
  for (int k0 = 0; k0 < K; k0 += 32) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int id = tid + it * 256, r = id >> 3, c = (id & 7) * 4;
      const int k = k0 + c;
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f), g = make_float4(1.f, 1.f, 1.f, 1.f), bq = a;
      if (vec && k + 3 < K) {
        if (xr[it]) a = *reinterpret_cast<const float4*>(xr[it] + k);
        if (gr[it]) g = *reinterpret_cast<const float4*>(gr[it] + k);
        if (wr[it]) bq = *reinterpret_cast<const float4*>(wr[it] + k);
      } else {
        float av[4] = {0.f, 0.f, 0.f, 0.f}, gv[4] = {1.f, 1.f, 1.f, 1.f}, bv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (k + u < K) {
            if (xr[it]) av[u] = xr[it][k + u];
            if (gr[it]) gv[u] = gr[it][k + u];
            if (wr[it]) bv[u] = wr[it][k + u];
          }
        a = make_float4(av[0], av[1], av[2], av[3]), g = make_float4(gv[0], gv[1], gv[2], gv[3]), bq = make_float4(bv[0], bv[1], bv[2], bv[3]);
      }
      if (gr[it]) a.x *= g.x, a.y *= g.y, a.z *= g.z, a.w *= g.w;
      sA[r][c] = a.x, sA[r][c + 1] = a.y, sA[r][c + 2] = a.z, sA[r][c + 3] = a.w;
      sB[r][c] = bq.x, sB[r][c + 1] = bq.y, sB[r][c + 2] = bq.z, sB[r][c + 3] = bq.w;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 32; kk += 2) {
      // D[i][j] += A[i][k] B[k][j] with i = output row m, j = output column n: a lane holds ONE column of 16 rows, so a store (and
      // the residual load) covers two 128-byte row segments instead of 64 rows x 4 bytes
      const float a = sA[wm + (lane & 31)][kk + (lane >> 5)];
      const float bq = sB[wn + (lane & 31)][kk + (lane >> 5)];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bq, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  const int n = n0 + wn + (lane & 31);
  if (n >= N) return;
  const float bv = bias != nullptr ? bias[n] : 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const long m = m0 + wm + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
    if (m >= M) continue;
    float v = acc[e] + bv;
    if (act) v = silu_(v);
    if (residual != nullptr) v += residual[m * N + n];
    y[m * N + n] = v;
  }
}

inline int grid_for(long n, int cap = 4096) {
  long b = (n + 255) / 256;
  if (b > cap) b = cap;
  return (int)(b < 1 ? 1 : b);
}

}  // namespace

extern "C" int aql_resize_bilinear_nhwc(const float* x, int B, int C, int H, int W, int Ho, int Wo, float* y,
                                        hipStream_t stream) {
  AQL_CHECK_ARG(x && y && B > 0 && C > 0, "aql_resize_bilinear_nhwc: bad args");
  hipLaunchKernelGGL(resize_bilinear_kernel, dim3(grid_for((long)B * Ho * Wo)), dim3(256), 0, stream, x, B, C, H, W, Ho,
                     Wo, y);
  AQL_CHECK_LAUNCH("aql_resize_bilinear_nhwc");
  return AQL_OK;
}
extern "C" int aql_stem_conv3x3s2_silu(const float* x, const float* w, const float* bias, int B, int H, int W, int Cout,
                                       float* y, hipStream_t stream) {
  AQL_CHECK_ARG(x && w && bias && y && Cout % 4 == 0 && Cout <= 64 && H % 2 == 0 && W % 2 == 0,
                "aql_stem_conv3x3s2_silu: bad args");
  hipLaunchKernelGGL(stem_conv_kernel, dim3(grid_for((long)B * (H / 2) * (W / 2) * (Cout / 4))), dim3(256), 0, stream, x,
                     w, bias, B, H, W, Cout, y);
  AQL_CHECK_LAUNCH("aql_stem_conv3x3s2_silu");
  return AQL_OK;
}
extern "C" int aql_dwconv_silu(const float* x, const float* w, const float* bias, int B, int H, int W, int C, int k,
                               int stride, float* y, hipStream_t stream) {
  AQL_CHECK_ARG(x && w && bias && y && C % 4 == 0 && (k == 3 || k == 5) && (stride == 1 || stride == 2),
                "aql_dwconv_silu: bad args");
  const int Ho = (H + 2 * (k / 2) - k) / stride + 1, Wo = (W + 2 * (k / 2) - k) / stride + 1;
  hipLaunchKernelGGL(dwconv_kernel, dim3(grid_for((long)B * Ho * Wo * (C / 4))), dim3(256), 0, stream, x, w, bias, B, H,
                     W, C, k, stride, y);
  AQL_CHECK_LAUNCH("aql_dwconv_silu");
  return AQL_OK;
}
extern "C" int aql_avgpool_nhwc(const float* x, int B, int HW, int C, float* out, hipStream_t stream) {
  AQL_CHECK_ARG(x && out, "aql_avgpool_nhwc: bad args");
  hipLaunchKernelGGL(avgpool_kernel, dim3((C + 63) / 64, B), dim3(256), 0, stream, x, HW, C, out);
  AQL_CHECK_LAUNCH("aql_avgpool_nhwc");
  return AQL_OK;
}
extern "C" int aql_se_gate(const float* pool, const float* w1, const float* b1, const float* w2, const float* b2, int B,
                           int C, int Cs, float* gate, hipStream_t stream) {
  AQL_CHECK_ARG(pool && w1 && b1 && w2 && b2 && gate && Cs <= 512 && C <= 2048, "aql_se_gate: bad args");
  hipLaunchKernelGGL(se_fc_kernel, dim3(B), dim3(1024), 0, stream, pool, 0, 1.f, w1, b1, w2, b2, C, Cs, gate);
  AQL_CHECK_LAUNCH("aql_se_gate");
  return AQL_OK;
}
// The pool of a squeeze-excite block as S pixel slabs: part [B][S][C] raw sums (aql_avgpool_nhwc_slabs), consumed by
// aql_se_gate_slabs, which forms the mean (sum over the slabs in order / HW) before the two small linears.  utils/models.py:84-96
// through torchvision's MBConv SqueezeExcitation (AdaptiveAvgPool2d(1) -> fc1 -> SiLU -> fc2 -> Sigmoid).
extern "C" int aql_avgpool_nhwc_slabs(const float* x, int B, int HW, int C, int S, float* part, hipStream_t stream) {
  AQL_CHECK_ARG(x && part && B > 0 && HW > 0 && C > 0 && S > 0 && S <= HW && B < 65536 && S < 65536, "aql_avgpool_nhwc_slabs: bad args");
  hipLaunchKernelGGL(avgpool_slabs_kernel, dim3((C + 63) / 64, B, S), dim3(256), 0, stream, x, HW, C, S, part);
  AQL_CHECK_LAUNCH("aql_avgpool_nhwc_slabs");
  return AQL_OK;
}
extern "C" int aql_se_gate_slabs(const float* part, int S, int HW, const float* w1, const float* b1, const float* w2, const float* b2,
                                 int B, int C, int Cs, float* gate, hipStream_t stream) {
  AQL_CHECK_ARG(part && w1 && b1 && w2 && b2 && gate && Cs <= 512 && C <= 2048 && S > 0 && HW > 0, "aql_se_gate_slabs: bad args");
  hipLaunchKernelGGL(se_fc_kernel, dim3(B), dim3(1024), 0, stream, part, S, 1.f / (float)HW, w1, b1, w2, b2, C, Cs, gate);
  AQL_CHECK_LAUNCH("aql_se_gate_slabs");
  return AQL_OK;
}
extern "C" int aql_pwconv_f32(const float* x, const float* w, const float* bias, const float* gate, int rows_per_sample,
                              const float* residual, long M, int N, int K, int act, float* y, hipStream_t stream) {
  AQL_CHECK_ARG(x && w && y && M > 0 && N > 0 && K > 0 && (gate == nullptr || rows_per_sample > 0),
                "aql_pwconv_f32: bad args");
  hipLaunchKernelGGL(pwconv_kernel, dim3((unsigned)((M + 63) / 64), (N + 63) / 64), dim3(256), 0, stream, x, w, bias,
                     gate, rows_per_sample > 0 ? rows_per_sample : 1, residual, M, N, K, act, y);
  AQL_CHECK_LAUNCH("aql_pwconv_f32");
  return AQL_OK;
}
