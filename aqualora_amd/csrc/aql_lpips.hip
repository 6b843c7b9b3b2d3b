// LPIPS(VGG16) pieces around the implicit-GEMM 3x3 convolutions (csrc/aql_gemm.hip) -- the perceptual loss of stage 1,
// `loss_fn_vgg = lpips.LPIPS(net='vgg')` at train/latent_wm_pretrain.py:111 and `loss_fn_vgg(clean_image, watermarked_image)`
// at :182.  lpips 0.1.4 is not on disk; its published algorithm (Zhang et al., "The Unreasonable Effectiveness of Deep
// Features as a Perceptual Metric", v0.1 linear heads):
//   x <- (x - shift) / scale  per RGB channel                                (ScalingLayer)
//   f_l = VGG16 features after relu1_2, relu2_2, relu3_3, relu4_3, relu5_3   (13 conv3x3 + ReLU, 4 max-pools)
//   u_l = f_l / (||f_l||_2 over channels + 1e-10)                            (normalize_tensor)
//   d   = sum_l  mean_{h,w}  sum_c  w_lc * (u0_lc - u1_lc)^2                 (1x1 "lin" heads, weights >= 0, no bias)
// All HBM-bound elementwise / per-pixel work: 16-byte accesses, channels-last bf16 activations, fp32 arithmetic.
#include "aql_common.h"

namespace {

__device__ __forceinline__ float bf16lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

inline int grid_for(long n, int cap = 8192) {
  long b = (n + 255) / 256;
  if (b > cap) b = cap;
  return (int)(b < 1 ? 1 : b);
}

// x: fp32 NCHW [B,3,H,W] in [-1,1]  ->  y: bf16 NHWC [B,H,W,8] (channels 3..7 zero: the conv kernels want Cin % 8 == 0)
__global__ __launch_bounds__(256) void lpips_scale_fwd_kernel(const float* __restrict__ x, long B, long HW, bf16_t* __restrict__ y) {
  const float sh[3] = {-0.030f, -0.088f, -0.188f}, sc[3] = {0.458f, 0.448f, 0.450f};
  const long n = B * HW;
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < n; id += (long)gridDim.x * blockDim.x) {
    const long b = id / HW, p = id - b * HW;
    float v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] = (x[(b * 3 + c) * HW + p] - sh[c]) / sc[c];
    *reinterpret_cast<uint4*>(y + id * 8) = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], 0.f), 0u, 0u);
  }
}

// dy: bf16 NHWC [B,H,W,8] -> dx: fp32 NCHW [B,3,H,W]
__global__ __launch_bounds__(256) void lpips_scale_bwd_kernel(const bf16_t* __restrict__ dy, long B, long HW, float* __restrict__ dx) {
  const float sc[3] = {0.458f, 0.448f, 0.450f};
  const long n = B * HW;
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < n; id += (long)gridDim.x * blockDim.x) {
    const long b = id / HW, p = id - b * HW;
    const uint2 g = *reinterpret_cast<const uint2*>(dy + id * 8);
    dx[(b * 3 + 0) * HW + p] = bf16lo(g.x) / sc[0];
    dx[(b * 3 + 1) * HW + p] = bf16hi(g.x) / sc[1];
    dx[(b * 3 + 2) * HW + p] = bf16lo(g.y) / sc[2];
  }
}

__device__ __forceinline__ uint32_t relu2(uint32_t w) {  // two packed bf16: zero the negative ones (sign bit set)
  return w & ~(((w & 0x8000u) ? 0xffffu : 0u) | ((w & 0x80000000u) ? 0xffff0000u : 0u));
}

// y = relu(x) elementwise, n8 chunks of 8 bf16
__global__ __launch_bounds__(256) void relu_fwd_kernel(const uint4* __restrict__ x, long n8, uint4* __restrict__ y) {
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < n8; id += (long)gridDim.x * blockDim.x) {
    const uint4 v = x[id];
    y[id] = make_uint4(relu2(v.x), relu2(v.y), relu2(v.z), relu2(v.w));
  }
}

__device__ __forceinline__ uint32_t gate2(uint32_t g, uint32_t y) {  // keep the gradient where the forward output is > 0
  return g & (((y & 0x7fffu) && !(y & 0x8000u) ? 0xffffu : 0u) | ((y & 0x7fff0000u) && !(y & 0x80000000u) ? 0xffff0000u : 0u));
}

// dx = dy * (y > 0)
__global__ __launch_bounds__(256) void relu_bwd_kernel(const uint4* __restrict__ dy, const uint4* __restrict__ y, long n8,
                                                       uint4* __restrict__ dx) {
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < n8; id += (long)gridDim.x * blockDim.x) {
    const uint4 g = dy[id], o = y[id];
    dx[id] = make_uint4(gate2(g.x, o.x), gate2(g.y, o.y), gate2(g.z, o.z), gate2(g.w, o.w));
  }
}

__device__ __forceinline__ uint32_t max2(uint32_t a, uint32_t b) {
  const float lo = fmaxf(bf16lo(a), bf16lo(b)), hi = fmaxf(bf16hi(a), bf16hi(b));
  return (__float_as_uint(lo) >> 16) | (__float_as_uint(hi) & 0xffff0000u);
}

// 2x2 / stride-2 max-pool on NHWC bf16 (H, W even), 8 channels per thread
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const bf16_t* __restrict__ x, long B, int H, int W, int C,
                                                          bf16_t* __restrict__ y) {
  const int Ho = H / 2, Wo = W / 2, C8 = C / 8;
  const long n = B * Ho * Wo * C8;
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < n; id += (long)gridDim.x * blockDim.x) {
    const int c = (int)(id % C8) * 8;
    long p = id / C8;
    const int wo = (int)(p % Wo);
    p /= Wo;
    const int ho = (int)(p % Ho);
    const long b = p / Ho;
    const bf16_t* s = x + ((b * H + 2 * ho) * W + 2 * wo) * (long)C + c;
    const uint4 a = *reinterpret_cast<const uint4*>(s), bb = *reinterpret_cast<const uint4*>(s + C);
    const uint4 cc = *reinterpret_cast<const uint4*>(s + (long)W * C), d = *reinterpret_cast<const uint4*>(s + (long)W * C + C);
    *reinterpret_cast<uint4*>(y + id * 8) = make_uint4(max2(max2(a.x, bb.x), max2(cc.x, d.x)), max2(max2(a.y, bb.y), max2(cc.y, d.y)),
                                                       max2(max2(a.z, bb.z), max2(cc.z, d.z)), max2(max2(a.w, bb.w), max2(cc.w, d.w)));
  }
}

// dx[window position] = dy where that position holds the window's max (first such position in raster order, like torch)
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ y,
                                                          const bf16_t* __restrict__ dy, long B, int H, int W, int C,
                                                          bf16_t* __restrict__ dx) {
  const int Ho = H / 2, Wo = W / 2, C8 = C / 8;
  const long n = B * Ho * Wo * C8;
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < n; id += (long)gridDim.x * blockDim.x) {
    const int c = (int)(id % C8) * 8;
    long p = id / C8;
    const int wo = (int)(p % Wo);
    p /= Wo;
    const int ho = (int)(p % Ho);
    const long b = p / Ho;
    const long base = ((b * H + 2 * ho) * W + 2 * wo) * (long)C + c;
    const long off[4] = {0, C, (long)W * C, (long)W * C + C};
    bf16_t m[8], g[8];
    *reinterpret_cast<uint4*>(m) = *reinterpret_cast<const uint4*>(y + id * 8);
    *reinterpret_cast<uint4*>(g) = *reinterpret_cast<const uint4*>(dy + id * 8);
    bool done[8] = {false, false, false, false, false, false, false, false};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      bf16_t v[8], o[8];
      *reinterpret_cast<uint4*>(v) = *reinterpret_cast<const uint4*>(x + base + off[q]);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const bool hit = !done[e] && v[e] == m[e];
        o[e] = hit ? g[e] : (bf16_t)0;
        done[e] |= hit;
      }
      *reinterpret_cast<uint4*>(dx + base + off[q]) = *reinterpret_cast<const uint4*>(o);
    }
  }
}

// One wavefront per pixel; f0, f1: post-ReLU features [B*HW][C] bf16 (f0 = reference image, f1 = the image that carries
// gradient).  out[b] += (1/HW) * sum_pixels sum_c w_c (f0_c/n0 - f1_c/n1)^2,  n = ||f||_2 + 1e-10.
// MODE 0: forward (atomicAdd of the per-workgroup partial sums).  MODE 1: backward, df1 = d(out[b]) / d(f1) * gout[b].
template <int MODE>
__global__ __launch_bounds__(256) void lpips_dist_kernel(const bf16_t* __restrict__ f0, const bf16_t* __restrict__ f1,
                                                         const float* __restrict__ w, long B, long HW, int C,
                                                         float* __restrict__ out, const float* __restrict__ gout,
                                                         bf16_t* __restrict__ df1) {
  __shared__ float part[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long npix = B * HW;
  const long b_blk = ((long)blockIdx.x * 4) / HW;   // HW % 4 == 0 is required by the host: a workgroup's 4 pixels share b
  float acc = 0.f;
  for (long pix = (long)blockIdx.x * 4 + wave; pix < npix; pix += (long)gridDim.x * 4) {
    const bf16_t* p0 = f0 + pix * C;
    const bf16_t* p1 = f1 + pix * C;
    float s0 = 0.f, s1 = 0.f;
    for (int c = lane * 8; c < C; c += 512) {
      const uint4 a = *reinterpret_cast<const uint4*>(p0 + c), bq = *reinterpret_cast<const uint4*>(p1 + c);
      const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {bq.x, bq.y, bq.z, bq.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s0 += bf16lo(aw[e]) * bf16lo(aw[e]) + bf16hi(aw[e]) * bf16hi(aw[e]);
        s1 += bf16lo(bw[e]) * bf16lo(bw[e]) + bf16hi(bw[e]) * bf16hi(bw[e]);
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      s0 += __shfl_xor(s0, o, 64);
      s1 += __shfl_xor(s1, o, 64);
    }
    const float r1 = sqrtf(s1), n0 = sqrtf(s0) + 1e-10f, n1 = r1 + 1e-10f;
    const float i0 = 1.f / n0, i1 = 1.f / n1;
    float d = 0.f, gf = 0.f;   // d = sum w (a-u)^2 ; gf = sum_c G_c f1_c  with G_c = -2 w_c (a_c - u_c)
    for (int c = lane * 8; c < C; c += 512) {
      const uint4 a = *reinterpret_cast<const uint4*>(p0 + c), bq = *reinterpret_cast<const uint4*>(p1 + c);
      const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {bq.x, bq.y, bq.z, bq.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float wl = w[c + 2 * e], wh = w[c + 2 * e + 1];
        const float fl = bf16lo(bw[e]), fh = bf16hi(bw[e]);
        const float dl = bf16lo(aw[e]) * i0 - fl * i1, dh = bf16hi(aw[e]) * i0 - fh * i1;
        d += wl * dl * dl + wh * dh * dh;
        gf += -2.f * (wl * dl * fl + wh * dh * fh);
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      d += __shfl_xor(d, o, 64);
      gf += __shfl_xor(gf, o, 64);
    }
    if (MODE == 0) {
      acc += d;
    } else {
      const long b = pix / HW;
      const float g = gout[b] / (float)HW;
      const float k2 = (r1 > 0.f) ? gf / (r1 * n1 * n1) : 0.f;
      for (int c = lane * 8; c < C; c += 512) {
        const uint4 a = *reinterpret_cast<const uint4*>(p0 + c), bq = *reinterpret_cast<const uint4*>(p1 + c);
        const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {bq.x, bq.y, bq.z, bq.w};
        uint32_t ow[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float wl = w[c + 2 * e], wh = w[c + 2 * e + 1];
          const float fl = bf16lo(bw[e]), fh = bf16hi(bw[e]);
          const float Gl = -2.f * wl * (bf16lo(aw[e]) * i0 - fl * i1), Gh = -2.f * wh * (bf16hi(aw[e]) * i0 - fh * i1);
          ow[e] = pack_bf16x2(g * (Gl * i1 - fl * k2), g * (Gh * i1 - fh * k2));
        }
        *reinterpret_cast<uint4*>(df1 + pix * C + c) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
      }
    }
  }
  if (MODE == 0) {
    if (lane == 0) part[wave] = acc;
    __syncthreads();
    // grid-stride: a workgroup may visit several samples only when gridDim.x * 4 < npix; the host launches one pass
    if (threadIdx.x == 0) atomicAdd(out + b_blk, (part[0] + part[1] + part[2] + part[3]) / (float)HW);
  }
}

}  // namespace

extern "C" int aql_lpips_scale(const float* x, int B, int H, int W, bf16_t* y, hipStream_t stream) {
  AQL_CHECK_ARG(x && y && B > 0 && H > 0 && W > 0, "aql_lpips_scale: bad args");
  const long n = (long)B * H * W;
  hipLaunchKernelGGL(lpips_scale_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, stream, x, (long)B, (long)H * W, y);
  AQL_CHECK_LAUNCH("aql_lpips_scale");
  return AQL_OK;
}

extern "C" int aql_lpips_scale_bwd(const bf16_t* dy, int B, int H, int W, float* dx, hipStream_t stream) {
  AQL_CHECK_ARG(dy && dx && B > 0 && H > 0 && W > 0, "aql_lpips_scale_bwd: bad args");
  const long n = (long)B * H * W;
  hipLaunchKernelGGL(lpips_scale_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, stream, dy, (long)B, (long)H * W, dx);
  AQL_CHECK_LAUNCH("aql_lpips_scale_bwd");
  return AQL_OK;
}

extern "C" int aql_relu_bf16(const bf16_t* x, long n, bf16_t* y, hipStream_t stream) {
  AQL_CHECK_ARG(x && y && n > 0 && n % 8 == 0, "aql_relu_bf16: bad args");
  hipLaunchKernelGGL(relu_fwd_kernel, dim3(grid_for(n / 8)), dim3(256), 0, stream, reinterpret_cast<const uint4*>(x), n / 8,
                     reinterpret_cast<uint4*>(y));
  AQL_CHECK_LAUNCH("aql_relu_bf16");
  return AQL_OK;
}

extern "C" int aql_relu_bf16_bwd(const bf16_t* dy, const bf16_t* y, long n, bf16_t* dx, hipStream_t stream) {
  AQL_CHECK_ARG(dy && y && dx && n > 0 && n % 8 == 0, "aql_relu_bf16_bwd: bad args");
  hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid_for(n / 8)), dim3(256), 0, stream, reinterpret_cast<const uint4*>(dy),
                     reinterpret_cast<const uint4*>(y), n / 8, reinterpret_cast<uint4*>(dx));
  AQL_CHECK_LAUNCH("aql_relu_bf16_bwd");
  return AQL_OK;
}

extern "C" int aql_maxpool2x2_nhwc(const bf16_t* x, int B, int H, int W, int C, bf16_t* y, hipStream_t stream) {
  AQL_CHECK_ARG(x && y && B > 0 && H % 2 == 0 && W % 2 == 0 && C % 8 == 0, "aql_maxpool2x2_nhwc: bad shape");
  const long n = (long)B * (H / 2) * (W / 2) * (C / 8);
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, stream, x, (long)B, H, W, C, y);
  AQL_CHECK_LAUNCH("aql_maxpool2x2_nhwc");
  return AQL_OK;
}

extern "C" int aql_maxpool2x2_nhwc_bwd(const bf16_t* x, const bf16_t* y, const bf16_t* dy, int B, int H, int W, int C,
                                       bf16_t* dx, hipStream_t stream) {
  AQL_CHECK_ARG(x && y && dy && dx && B > 0 && H % 2 == 0 && W % 2 == 0 && C % 8 == 0, "aql_maxpool2x2_nhwc_bwd: bad shape");
  const long n = (long)B * (H / 2) * (W / 2) * (C / 8);
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, stream, x, y, dy, (long)B, H, W, C, dx);
  AQL_CHECK_LAUNCH("aql_maxpool2x2_nhwc_bwd");
  return AQL_OK;
}

// out[b] += mean_{h,w} sum_c w_c (f0/(||f0||+eps) - f1/(||f1||+eps))^2 ; out must be zeroed by the caller before the first
// layer (the five layers accumulate into it).  HW % 4 == 0.
extern "C" int aql_lpips_layer(const bf16_t* f0, const bf16_t* f1, const float* w, int B, long HW, int C, float* out,
                               hipStream_t stream) {
  AQL_CHECK_ARG(f0 && f1 && w && out && B > 0 && HW > 0 && HW % 4 == 0 && C % 8 == 0, "aql_lpips_layer: bad shape");
  const long blocks = (long)B * HW / 4;   // one pass: every workgroup stays inside one sample
  AQL_CHECK_ARG(blocks < (1L << 31), "aql_lpips_layer: too many pixels");
  hipLaunchKernelGGL(lpips_dist_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, stream, f0, f1, w, (long)B, HW, C, out, nullptr,
                     nullptr);
  AQL_CHECK_LAUNCH("aql_lpips_layer");
  return AQL_OK;
}

extern "C" int aql_lpips_layer_bwd(const bf16_t* f0, const bf16_t* f1, const float* w, int B, long HW, int C,
                                   const float* gout, bf16_t* df1, hipStream_t stream) {
  AQL_CHECK_ARG(f0 && f1 && w && gout && df1 && B > 0 && HW > 0 && HW % 4 == 0 && C % 8 == 0, "aql_lpips_layer_bwd: bad shape");
  const long blocks = (long)B * HW / 4;
  AQL_CHECK_ARG(blocks < (1L << 31), "aql_lpips_layer_bwd: too many pixels");
  hipLaunchKernelGGL(lpips_dist_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, stream, f0, f1, w, (long)B, HW, C, nullptr, gout,
                     df1);
  AQL_CHECK_LAUNCH("aql_lpips_layer_bwd");
  return AQL_OK;
}
