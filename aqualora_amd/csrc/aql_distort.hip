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

// one 1-D pass of the separable Gaussian (axis 0 = x, 1 = y); BWD applies the adjoint (scatter through the reflection).
// taps: k weights shared by every image (tap_stride 0) or one row of k weights per SAMPLE (tap_stride = k; kornia's
// RandomGaussianBlur draws one sigma per sample); C = channels per sample.
template <bool BWD>
__global__ __launch_bounds__(256) void blur1d_kernel(const float* __restrict__ src, float* __restrict__ dst, int BC, int H,
                                                     int W, int axis, int k, const float* __restrict__ taps, int C,
                                                     int tap_stride) {
  const long n = (long)BC * H * W;
  const int r = k / 2;
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < n; id += (long)gridDim.x * blockDim.x) {
    const int x = (int)(id % W), y = (int)((id / W) % H);
    const long img = id / ((long)W * H);
    const long base = img * H * W;
    const float* tp = taps + (img / C) * tap_stride;
    if (!BWD) {
      float a = 0.f;
      for (int t = 0; t < k; ++t) {
        const int xx = axis == 0 ? reflect(x + t - r, W) : x, yy = axis == 1 ? reflect(y + t - r, H) : y;
        a += tp[t] * src[base + (long)yy * W + xx];
      }
      dst[id] = a;
    } else {
      const float g = src[id];
      for (int t = 0; t < k; ++t) {
        const int xx = axis == 0 ? reflect(x + t - r, W) : x, yy = axis == 1 ? reflect(y + t - r, H) : y;
        atomicAdd(dst + base + (long)yy * W + xx, tp[t] * g);
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

// ---------------------------------------------------------------------------------------------------------------
// Colour jiggle (kornia 0.6.12 ColorJiggle as called at noises.py:97-103, noiser.py:52-57, utils_eval.py:271-276):
// brightness (additive shift, clamp), contrast (multiplicative, clamp), saturation and hue (through HSV), applied in a
// caller-supplied order.  One thread per pixel.  The backward pass evaluates the same code on dual numbers
// (value + gradient w.r.t. the pixel's r,g,b), so forward and backward cannot drift apart.
// ---------------------------------------------------------------------------------------------------------------
struct D3 {  // value + d/d(r_in, g_in, b_in)
  float v, d[3];
};
__device__ __forceinline__ D3 operator+(D3 a, D3 b) { return {a.v + b.v, {a.d[0] + b.d[0], a.d[1] + b.d[1], a.d[2] + b.d[2]}}; }
__device__ __forceinline__ D3 operator-(D3 a, D3 b) { return {a.v - b.v, {a.d[0] - b.d[0], a.d[1] - b.d[1], a.d[2] - b.d[2]}}; }
__device__ __forceinline__ D3 operator*(D3 a, D3 b) {
  return {a.v * b.v, {a.d[0] * b.v + a.v * b.d[0], a.d[1] * b.v + a.v * b.d[1], a.d[2] * b.v + a.v * b.d[2]}};
}
__device__ __forceinline__ D3 operator/(D3 a, D3 b) {
  const float q = a.v / b.v, ib = 1.f / b.v;
  return {q, {(a.d[0] - q * b.d[0]) * ib, (a.d[1] - q * b.d[1]) * ib, (a.d[2] - q * b.d[2]) * ib}};
}
__device__ __forceinline__ D3 operator+(D3 a, float b) { return {a.v + b, {a.d[0], a.d[1], a.d[2]}}; }
__device__ __forceinline__ D3 operator-(D3 a, float b) { return {a.v - b, {a.d[0], a.d[1], a.d[2]}}; }
__device__ __forceinline__ D3 operator*(D3 a, float b) { return {a.v * b, {a.d[0] * b, a.d[1] * b, a.d[2] * b}}; }
__device__ __forceinline__ D3 operator-(float a, D3 b) { return {a - b.v, {-b.d[0], -b.d[1], -b.d[2]}}; }
__device__ __forceinline__ float val(float a) { return a; }
__device__ __forceinline__ float val(D3 a) { return a.v; }
__device__ __forceinline__ float cst(float, float c) { return c; }      // a constant of the same scalar type
__device__ __forceinline__ D3 cst(D3, float c) { return {c, {0.f, 0.f, 0.f}}; }
__device__ __forceinline__ float with_val(float, float v) { return v; }  // same derivative, new value (mod / wrap)
__device__ __forceinline__ D3 with_val(D3 a, float v) { return {v, {a.d[0], a.d[1], a.d[2]}}; }

template <class T>
__device__ __forceinline__ T clamp01(T a) {
  const float v = val(a);
  return v < 0.f ? cst(a, 0.f) : (v > 1.f ? cst(a, 1.f) : a);
}

constexpr float TWO_PI = 6.283185307179586f;

template <class T>
__device__ __forceinline__ void rgb_to_hsv(T r, T g, T b, T& h, T& s, T& v) {
  // first maximum wins (torch.max index), like the gather on argmax in kornia.color.rgb_to_hsv
  int am = 0;
  T mx = r;
  if (val(g) > val(mx)) mx = g, am = 1;
  if (val(b) > val(mx)) mx = b, am = 2;
  T mn = r;
  if (val(g) < val(mn)) mn = g;
  if (val(b) < val(mn)) mn = b;
  T delta = mx - mn;
  v = mx;
  s = delta / (mx + 1e-8f);
  if (val(delta) == 0.f) delta = cst(delta, 1.f);
  const T rc = mx - r, gc = mx - g, bc = mx - b;
  T hh = am == 0 ? (bc - gc) : (am == 1 ? (rc - bc) + delta * 2.f : (gc - rc) + delta * 4.f);
  hh = hh / delta;
  hh = hh * (1.f / 6.f);
  const float w = val(hh) - floorf(val(hh));  // python-style % 1.0
  h = with_val(hh, w) * TWO_PI;
}

template <class T>
__device__ __forceinline__ void hsv_to_rgb(T h, T s, T v, T& r, T& g, T& b) {
  const T h6 = h * (6.f / TWO_PI);
  const float fl = floorf(val(h6));
  int hi = (int)fl % 6;
  if (hi < 0) hi += 6;
  const T f = with_val(h6, val(h6) - fl);  // (h*6) % 6 - hi  ==  frac(h*6)
  const T one = cst(v, 1.f);
  const T p = v * (one - s), q = v * (one - f * s), t = v * (one - (one - f) * s);
  switch (hi) {
    case 0: r = v, g = t, b = p; break;
    case 1: r = q, g = v, b = p; break;
    case 2: r = p, g = v, b = t; break;
    case 3: r = p, g = q, b = v; break;
    case 4: r = t, g = p, b = v; break;
    default: r = v, g = p, b = q; break;
  }
}

// fac = {brightness shift (factor - 1), contrast, saturation, hue shift in radians}; order[i] in 0..3 names the i-th op
template <class T>
__device__ __forceinline__ void jiggle(T& r, T& g, T& b, const float* fac, const int* order) {
#pragma unroll 1
  for (int i = 0; i < 4; ++i) {
    const int op = order[i];
    if (op == 0) {
      r = clamp01(r + fac[0]), g = clamp01(g + fac[0]), b = clamp01(b + fac[0]);
    } else if (op == 1) {
      r = clamp01(r * fac[1]), g = clamp01(g * fac[1]), b = clamp01(b * fac[1]);
    } else {
      T h, s, v;
      rgb_to_hsv(r, g, b, h, s, v);
      if (op == 2) {
        s = clamp01(s * fac[2]);
      } else {
        const T hs = h + fac[3];
        h = with_val(hs, fmodf(val(hs), TWO_PI));  // torch.fmod: sign of the dividend
      }
      hsv_to_rgb(h, s, v, r, g, b);
    }
  }
}

template <bool BWD>
__global__ __launch_bounds__(256) void color_jiggle_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           float* __restrict__ out, int B, long HW,
                                                           const float* __restrict__ factors,
                                                           const int* __restrict__ order) {
  int ord[4] = {order[0], order[1], order[2], order[3]};
  const long n = (long)B * HW;
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < n; id += (long)gridDim.x * blockDim.x) {
    const long bi = id / HW, p = id - bi * HW;
    const long base = bi * 3 * HW + p;
    const float fac[4] = {factors[bi * 4 + 0], factors[bi * 4 + 1], factors[bi * 4 + 2], factors[bi * 4 + 3]};
    if (!BWD) {
      float r = x[base], g = x[base + HW], b = x[base + 2 * HW];
      jiggle(r, g, b, fac, ord);
      out[base] = r, out[base + HW] = g, out[base + 2 * HW] = b;
    } else {
      D3 r = {x[base], {1.f, 0.f, 0.f}}, g = {x[base + HW], {0.f, 1.f, 0.f}}, b = {x[base + 2 * HW], {0.f, 0.f, 1.f}};
      jiggle(r, g, b, fac, ord);
      const float gr = dy[base], gg = dy[base + HW], gb = dy[base + 2 * HW];
#pragma unroll
      for (int c = 0; c < 3; ++c) out[base + c * HW] = gr * r.d[c] + gg * g.d[c] + gb * b.d[c];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Rotation about the image centre (kornia RandomRotation -> rotate -> warp_affine, bilinear, zeros padding,
// align_corners=True; noises.py:20-31, utils_eval.py:292).  Positive angle = anti-clockwise (OpenCV convention of
// get_rotation_matrix2d).  Output pixel (x,y) samples the source at R(x - c) + c.  BWD scatters the adjoint.
// ---------------------------------------------------------------------------------------------------------------
template <bool BWD>
__global__ __launch_bounds__(256) void rotate_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int C,
                                                     int H, int W, const float* __restrict__ angle_deg) {
  const long n = (long)B * C * H * W;
  const float cx = 0.5f * (W - 1), cy = 0.5f * (H - 1);
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < n; id += (long)gridDim.x * blockDim.x) {
    const int x = (int)(id % W), y = (int)((id / W) % H);
    const long bc = id / ((long)W * H);
    const float th = angle_deg[bc / C] * 0.017453292519943295f;
    const float cs = cosf(th), sn = sinf(th);
    const float sx = cs * (x - cx) - sn * (y - cy) + cx, sy = sn * (x - cx) + cs * (y - cy) + cy;
    const float fx = floorf(sx), fy = floorf(sy);
    const int x0 = (int)fx, y0 = (int)fy;
    const float wx = sx - fx, wy = sy - fy;
    const long base = bc * H * W;
    float acc = 0.f;
    const float g = BWD ? src[id] : 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int xx = x0 + i, yy = y0 + j;
        if (xx < 0 || xx >= W || yy < 0 || yy >= H) continue;
        const float w = (i ? wx : 1.f - wx) * (j ? wy : 1.f - wy);
        if (!BWD) acc += w * src[base + (long)yy * W + xx];
        else atomicAdd(dst + base + (long)yy * W + xx, g * w);
      }
    if (!BWD) dst[id] = acc;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Sharpness (kornia RandomSharpness -> sharpness(); noises.py:106-119, utils_eval.py:294): blend of the image with a
// 3x3 smoothed copy ([[1,1,1],[1,5,1],[1,1,1]]/13, valid region only, clamped to [0,1]; the 1-pixel border keeps the
// input):  out = blur + f * (x - blur), clamped to [0,1] unless 0 < f < 1.
//   fwd:  dst = out.     bwd pass 0: tmp = (1-f) * g_eff * [interior] * [0 < blur < 1],  dst = f_eff * g_eff
//                        bwd pass 1: dst += sum_taps w * tmp(neighbours)         (g_eff = dy masked by the final clamp)
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float smooth3(const float* p, int W) {
  return (p[-W - 1] + p[-W] + p[-W + 1] + p[-1] + 5.f * p[0] + p[1] + p[W - 1] + p[W] + p[W + 1]) * (1.f / 13.f);
}

template <int MODE>  // 0 = forward, 1 = backward pass 0, 2 = backward pass 1
__global__ __launch_bounds__(256) void sharpness_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                        float* __restrict__ dst, float* __restrict__ tmp, int B, int C,
                                                        int H, int W, const float* __restrict__ factor) {
  const long n = (long)B * C * H * W;
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < n; id += (long)gridDim.x * blockDim.x) {
    const int px = (int)(id % W), py = (int)((id / W) % H);
    const long bc = id / ((long)W * H);
    const float f = factor[bc / C];
    const bool interior = px > 0 && px < W - 1 && py > 0 && py < H - 1;
    if (MODE == 2) {
      float acc = 0.f;
      const float* t = tmp + id;
#pragma unroll
      for (int j = -1; j <= 1; ++j)
#pragma unroll
        for (int i = -1; i <= 1; ++i) {
          const int xx = px + i, yy = py + j;
          if (xx < 0 || xx >= W || yy < 0 || yy >= H) continue;
          acc += ((i == 0 && j == 0) ? 5.f : 1.f) * t[(long)j * W + i];
        }
      dst[id] += acc * (1.f / 13.f);
      continue;
    }
    const float xv = x[id];
    const float raw = interior ? smooth3(x + id, W) : xv;
    const float blur = interior ? fminf(fmaxf(raw, 0.f), 1.f) : xv;
    const float res = blur + (xv - blur) * f;
    const bool clampit = !(f > 0.f && f < 1.f) && f != 0.f && f != 1.f;
    if (MODE == 0) {
      dst[id] = f == 0.f ? blur : (f == 1.f ? xv : (clampit ? fminf(fmaxf(res, 0.f), 1.f) : res));
    } else {
      float g = dy[id];
      if (clampit && (res < 0.f || res > 1.f)) g = 0.f;
      const float fe = f == 0.f ? 0.f : (f == 1.f ? 1.f : f);
      // border pixels: blur == x, so d(out)/dx = 1
      dst[id] = interior ? fe * g : g;
      tmp[id] = (interior && raw > 0.f && raw < 1.f) ? (1.f - fe) * g : 0.f;
    }
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

// Separable Gaussian blur with reflect borders, kernel (ky, kx) -- kornia's RandomGaussianBlur((ky, kx), sigma range) as
// called at noises.py:68 ((3, 9)), noiser.py:63 ((3, 5)) and utils_eval.py:280 ((3, 3)).  taps_x: kx weights, taps_y: ky
// weights, shared (per_sample = 0) or [B][k] rows, one per sample (per_sample = 1).  tmp: scratch of B*C*H*W floats.
extern "C" int aql_gauss_blur2(const float* src, float* dst, float* tmp, int B, int C, int H, int W, int kx, int ky,
                               const float* taps_x, const float* taps_y, int per_sample, int backward, hipStream_t stream) {
  AQL_CHECK_ARG(src && dst && tmp && taps_x && taps_y && kx % 2 == 1 && ky % 2 == 1 && kx >= 1 && ky >= 1 && kx / 2 < W &&
                    ky / 2 < H && B > 0 && C > 0,
                "aql_gauss_blur2: bad args");
  const int BC = B * C;
  const long n = (long)BC * H * W;
  const int sx = per_sample ? kx : 0, sy = per_sample ? ky : 0;
  if (backward) {
    (void)hipMemsetAsync(tmp, 0, n * sizeof(float), stream);
    (void)hipMemsetAsync(dst, 0, n * sizeof(float), stream);
    hipLaunchKernelGGL(blur1d_kernel<true>, dim3(grid_for(n)), dim3(256), 0, stream, src, tmp, BC, H, W, 1, ky, taps_y, C, sy);
    hipLaunchKernelGGL(blur1d_kernel<true>, dim3(grid_for(n)), dim3(256), 0, stream, tmp, dst, BC, H, W, 0, kx, taps_x, C, sx);
  } else {
    hipLaunchKernelGGL(blur1d_kernel<false>, dim3(grid_for(n)), dim3(256), 0, stream, src, tmp, BC, H, W, 0, kx, taps_x, C, sx);
    hipLaunchKernelGGL(blur1d_kernel<false>, dim3(grid_for(n)), dim3(256), 0, stream, tmp, dst, BC, H, W, 1, ky, taps_y, C, sy);
  }
  AQL_CHECK_LAUNCH("aql_gauss_blur2");
  return AQL_OK;
}

// square kernel, one tap set for the whole batch (kept for callers of the round-1 ABI)
extern "C" int aql_gauss_blur(const float* src, float* dst, float* tmp, int BC, int H, int W, int k, const float* taps,
                              int backward, hipStream_t stream) {
  AQL_CHECK_ARG(src && dst && tmp && taps && k % 2 == 1 && k >= 1 && k / 2 < H && k / 2 < W, "aql_gauss_blur: bad args");
  return aql_gauss_blur2(src, dst, tmp, BC, 1, H, W, k, k, taps, taps, 0, backward, stream);
}

extern "C" int aql_add_gauss_noise(const float* x, const float* noise, float std, int clamp01, long n, float* y,
                                   hipStream_t stream) {
  AQL_CHECK_ARG(x && noise && y, "aql_add_gauss_noise: bad args");
  hipLaunchKernelGGL(add_noise_clamp_kernel, dim3(grid_for(n)), dim3(256), 0, stream, x, noise, std, clamp01, n, y);
  AQL_CHECK_LAUNCH("aql_add_gauss_noise");
  return AQL_OK;
}

// x, y: [B,3,H,W] fp32 in [0,1]; factors: device [B][4] = {brightness shift, contrast, saturation, hue shift (rad)};
// order: device int[4], a permutation of {0:brightness, 1:contrast, 2:saturation, 3:hue}.  dy == null: y = jiggle(x);
// else y receives dx = J(x)^T dy.
extern "C" int aql_color_jiggle(const float* x, const float* dy, float* y, int B, int H, int W, const float* factors,
                                const int* order, hipStream_t stream) {
  AQL_CHECK_ARG(x && y && factors && order && B > 0 && H > 0 && W > 0, "aql_color_jiggle: bad args");
  const long n = (long)B * H * W;
  if (dy) hipLaunchKernelGGL(color_jiggle_kernel<true>, dim3(grid_for(n)), dim3(256), 0, stream, x, dy, y, B, (long)H * W, factors, order);
  else hipLaunchKernelGGL(color_jiggle_kernel<false>, dim3(grid_for(n)), dim3(256), 0, stream, x, dy, y, B, (long)H * W, factors, order);
  AQL_CHECK_LAUNCH("aql_color_jiggle");
  return AQL_OK;
}

// src/dst: [B,C,H,W] fp32; angle_deg: device [B] (anti-clockwise).  backward = 1: src is dy, dst receives dx.
extern "C" int aql_rotate_bilinear(const float* src, float* dst, int B, int C, int H, int W, const float* angle_deg,
                                   int backward, hipStream_t stream) {
  AQL_CHECK_ARG(src && dst && angle_deg && B > 0 && C > 0 && H > 1 && W > 1, "aql_rotate_bilinear: bad args");
  const long n = (long)B * C * H * W;
  if (backward) {
    (void)hipMemsetAsync(dst, 0, n * sizeof(float), stream);
    hipLaunchKernelGGL(rotate_kernel<true>, dim3(grid_for(n)), dim3(256), 0, stream, src, dst, B, C, H, W, angle_deg);
  } else {
    hipLaunchKernelGGL(rotate_kernel<false>, dim3(grid_for(n)), dim3(256), 0, stream, src, dst, B, C, H, W, angle_deg);
  }
  AQL_CHECK_LAUNCH("aql_rotate_bilinear");
  return AQL_OK;
}

// x: [B,C,H,W] fp32 in [0,1]; factor: device [B].  dy == null: dst = sharpness(x).  Else dst = dx (tmp: B*C*H*W floats).
extern "C" int aql_sharpness(const float* x, const float* dy, float* dst, float* tmp, int B, int C, int H, int W,
                             const float* factor, hipStream_t stream) {
  AQL_CHECK_ARG(x && dst && factor && B > 0 && C > 0 && H > 2 && W > 2 && (dy == nullptr || tmp != nullptr),
                "aql_sharpness: bad args");
  const long n = (long)B * C * H * W;
  if (!dy) {
    hipLaunchKernelGGL(sharpness_kernel<0>, dim3(grid_for(n)), dim3(256), 0, stream, x, dy, dst, tmp, B, C, H, W, factor);
  } else {
    hipLaunchKernelGGL(sharpness_kernel<1>, dim3(grid_for(n)), dim3(256), 0, stream, x, dy, dst, tmp, B, C, H, W, factor);
    hipLaunchKernelGGL(sharpness_kernel<2>, dim3(grid_for(n)), dim3(256), 0, stream, x, dy, dst, tmp, B, C, H, W, factor);
  }
  AQL_CHECK_LAUNCH("aql_sharpness");
  return AQL_OK;
}
