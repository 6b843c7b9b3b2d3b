// Image-space distortions of the robustness pipeline (reference utils/noise_layers/noises.py:34-85, noiser.py:46-71),
// NCHW fp32 like the reference.  The random parameters (crop window, sizes, kernel size, sigma, std) are drawn by the
// host exactly where the reference draws them (numpy RNG); the kernels are the deterministic image maps and their
// adjoints: crop + bilinear resize (torchvision Resize, antialias=None == F.interpolate(bilinear, align_corners=False)),
// separable Gaussian blur with reflect borders (kornia RandomGaussianBlur, recalled), additive Gaussian noise (+clamp).
#include "aql_common.h"

namespace {

__device__ __forceinline__ void src_coord(int o, float scale, int n, int& i0, int& i1, float& w) {
  const float f = fmaxf((o + 0.5f) * scale - 0.5f, 0.f);
  i0 = min((int)f, n - 1);
  i1 = min(i0 + 1, n - 1);
  w = f - i0;
}

// y[b,c,oy,ox] = bilinear sample of the crop window (top,left,ch,cw) of x; backward scatters with atomics
template <bool BWD>
__global__ __launch_bounds__(256) void crop_resize_kernel(const float* __restrict__ src, float* __restrict__ dst, int BC,
                                                          int H, int W, int top, int left, int ch, int cw, int oh,
                                                          int ow) {
  const long n = (long)BC * oh * ow;
  const float sh = (float)ch / oh, sw = (float)cw / ow;
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < n; id += (long)gridDim.x * blockDim.x) {
    const int ox = (int)(id % ow), oy = (int)((id / ow) % oh);
    const long bc = id / ((long)ow * oh);
    int y0, y1, x0, x1;
    float wy, wx;
    src_coord(oy, sh, ch, y0, y1, wy);
    src_coord(ox, sw, cw, x0, x1, wx);
    const long base = bc * H * W;
    const long p00 = base + (long)(top + y0) * W + left + x0, p01 = base + (long)(top + y0) * W + left + x1;
    const long p10 = base + (long)(top + y1) * W + left + x0, p11 = base + (long)(top + y1) * W + left + x1;
    if (!BWD) {
      dst[id] = (1.f - wy) * ((1.f - wx) * src[p00] + wx * src[p01]) + wy * ((1.f - wx) * src[p10] + wx * src[p11]);
    } else {
      const float g = src[id];  // src = dy [BC,oh,ow], dst = dx [BC,H,W] (pre-zeroed)
      atomicAdd(dst + p00, g * (1.f - wy) * (1.f - wx));
      atomicAdd(dst + p01, g * (1.f - wy) * wx);
      atomicAdd(dst + p10, g * wy * (1.f - wx));
      atomicAdd(dst + p11, g * wy * wx);
    }
  }
}

__device__ __forceinline__ int reflect(int i, int n) {  // 'reflect' padding: -1 -> 1, n -> n-2
  if (i < 0) i = -i;
  if (i >= n) i = 2 * n - 2 - i;
  return i;
}

// one 1-D pass of the separable Gaussian (axis 0 = x, 1 = y); BWD applies the adjoint (scatter through the reflection)
template <bool BWD>
__global__ __launch_bounds__(256) void blur1d_kernel(const float* __restrict__ src, float* __restrict__ dst, int BC, int H,
                                                     int W, int axis, int k, const float* __restrict__ taps) {
  const long n = (long)BC * H * W;
  const int r = k / 2;
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < n; id += (long)gridDim.x * blockDim.x) {
    const int x = (int)(id % W), y = (int)((id / W) % H);
    const long base = (id / ((long)W * H)) * H * W;
    if (!BWD) {
      float a = 0.f;
      for (int t = 0; t < k; ++t) {
        const int xx = axis == 0 ? reflect(x + t - r, W) : x, yy = axis == 1 ? reflect(y + t - r, H) : y;
        a += taps[t] * src[base + (long)yy * W + xx];
      }
      dst[id] = a;
    } else {
      const float g = src[id];
      for (int t = 0; t < k; ++t) {
        const int xx = axis == 0 ? reflect(x + t - r, W) : x, yy = axis == 1 ? reflect(y + t - r, H) : y;
        atomicAdd(dst + base + (long)yy * W + xx, taps[t] * g);
      }
    }
  }
}

__global__ __launch_bounds__(256) void add_noise_clamp_kernel(const float* __restrict__ x, const float* __restrict__ nz,
                                                              float std, int clamp01, long n, float* __restrict__ y) {
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < n; id += (long)gridDim.x * blockDim.x) {
    float v = x[id] + std * nz[id];
    if (clamp01) v = fminf(fmaxf(v, 0.f), 1.f);
    y[id] = v;
  }
}

inline int grid_for(long n) {
  long b = (n + 255) / 256;
  return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" int aql_crop_resize_bilinear(const float* src, float* dst, int BC, int H, int W, int top, int left, int ch,
                                        int cw, int oh, int ow, int backward, hipStream_t stream) {
  AQL_CHECK_ARG(src && dst && top >= 0 && left >= 0 && top + ch <= H && left + cw <= W && ch > 0 && cw > 0 && oh > 0 &&
                    ow > 0,
                "aql_crop_resize_bilinear: bad window");
  const long n = (long)BC * oh * ow;
  if (backward) {
    (void)hipMemsetAsync(dst, 0, (size_t)BC * H * W * sizeof(float), stream);
    hipLaunchKernelGGL(crop_resize_kernel<true>, dim3(grid_for(n)), dim3(256), 0, stream, src, dst, BC, H, W, top, left,
                       ch, cw, oh, ow);
  } else {
    hipLaunchKernelGGL(crop_resize_kernel<false>, dim3(grid_for(n)), dim3(256), 0, stream, src, dst, BC, H, W, top, left,
                       ch, cw, oh, ow);
  }
  AQL_CHECK_LAUNCH("aql_crop_resize_bilinear");
  return AQL_OK;
}

// taps: k normalised Gaussian weights on the device; tmp: scratch of BC*H*W floats
extern "C" int aql_gauss_blur(const float* src, float* dst, float* tmp, int BC, int H, int W, int k, const float* taps,
                              int backward, hipStream_t stream) {
  AQL_CHECK_ARG(src && dst && tmp && taps && k % 2 == 1 && k >= 1 && k / 2 < H && k / 2 < W, "aql_gauss_blur: bad args");
  const long n = (long)BC * H * W;
  if (backward) {
    (void)hipMemsetAsync(tmp, 0, n * sizeof(float), stream);
    (void)hipMemsetAsync(dst, 0, n * sizeof(float), stream);
    hipLaunchKernelGGL(blur1d_kernel<true>, dim3(grid_for(n)), dim3(256), 0, stream, src, tmp, BC, H, W, 1, k, taps);
    hipLaunchKernelGGL(blur1d_kernel<true>, dim3(grid_for(n)), dim3(256), 0, stream, tmp, dst, BC, H, W, 0, k, taps);
  } else {
    hipLaunchKernelGGL(blur1d_kernel<false>, dim3(grid_for(n)), dim3(256), 0, stream, src, tmp, BC, H, W, 0, k, taps);
    hipLaunchKernelGGL(blur1d_kernel<false>, dim3(grid_for(n)), dim3(256), 0, stream, tmp, dst, BC, H, W, 1, k, taps);
  }
  AQL_CHECK_LAUNCH("aql_gauss_blur");
  return AQL_OK;
}

extern "C" int aql_add_gauss_noise(const float* x, const float* noise, float std, int clamp01, long n, float* y,
                                   hipStream_t stream) {
  AQL_CHECK_ARG(x && noise && y, "aql_add_gauss_noise: bad args");
  hipLaunchKernelGGL(add_noise_clamp_kernel, dim3(grid_for(n)), dim3(256), 0, stream, x, noise, std, clamp01, n, y);
  AQL_CHECK_LAUNCH("aql_add_gauss_noise");
  return AQL_OK;
}
