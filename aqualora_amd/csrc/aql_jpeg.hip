// Differentiable JPEG simulation (reference utils/noise_layers/jpeg_compression.py:67-162): per 8x8 block
// RGB->YUV, 2-D DCT, keep the first 25/9/9 zig-zag coefficients of Y/U/V, inverse DCT, YUV->RGB.  The whole layer is a
// fixed linear map per 8x8x3 block, so ONE kernel serves forward and backward: out = Clast . Q (mask o (P X P^T)) Q^T
// with (P,Q,Cfirst,Clast) = (DCT, IDCT, rgb2yuv, yuv2rgb) forward and (IDCT^T, DCT^T, yuv2rgb^T, rgb2yuv^T) backward.
// NCHW fp32 in/out like the reference; images whose sides are not multiples of 8 are zero-padded and cropped (:133-160).
#include "aql_common.h"

namespace {

struct JpegParams {
  float P[64], Q[64];      // 8x8 row-major
  float c_first[9], c_last[9];
  unsigned long long mask[3];  // bit (ky*8+kx) set = coefficient kept
};

__global__ __launch_bounds__(192) void jpeg_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int H,
                                                   int W, const JpegParams prm) {
  __shared__ float sP[64], sQ[64];
  __shared__ float sblk[64][3][65];
  const int tid = threadIdx.x;
  if (tid < 64) {
    sP[tid] = prm.P[tid];
    sQ[tid] = prm.Q[tid];
  }
  __syncthreads();
  const int bw = (W + 7) / 8, bh = (H + 7) / 8;
  const long nblk = (long)B * bh * bw;
  const int lb = tid / 3, c = tid - lb * 3;  // 64 blocks x 3 channels per workgroup
  const long blk = (long)blockIdx.x * 64 + lb;
  const bool live = blk < nblk;
  const int bx = live ? (int)(blk % bw) : 0;
  const int by = live ? (int)((blk / bw) % bh) : 0;
  const int b = live ? (int)(blk / ((long)bw * bh)) : 0;
  float v[64];
  // first colour transform, channel c of this block (zero padding outside the image)
  const float c0 = prm.c_first[c * 3 + 0], c1 = prm.c_first[c * 3 + 1], c2 = prm.c_first[c * 3 + 2];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int yy = by * 8 + i;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int xx = bx * 8 + j;
      float r = 0.f;
      if (live && yy < H && xx < W) {
        const long o = ((long)b * 3 * H + yy) * W + xx;
        r = c0 * x[o] + c1 * x[o + (long)H * W] + c2 * x[o + 2L * H * W];
      }
      v[i * 8 + j] = r;
    }
  }
  // coef = P V P^T
  float t[64];
#pragma unroll
  for (int k = 0; k < 8; ++k)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float a = 0.f;
#pragma unroll
      for (int n = 0; n < 8; ++n) a += sP[k * 8 + n] * v[n * 8 + j];
      t[k * 8 + j] = a;
    }
  const unsigned long long m = prm.mask[c];
#pragma unroll
  for (int k = 0; k < 8; ++k)
#pragma unroll
    for (int l = 0; l < 8; ++l) {
      float a = 0.f;
#pragma unroll
      for (int n = 0; n < 8; ++n) a += t[k * 8 + n] * sP[l * 8 + n];
      v[k * 8 + l] = ((m >> (k * 8 + l)) & 1ull) ? a : 0.f;
    }
  // out = Q V Q^T
#pragma unroll
  for (int k = 0; k < 8; ++k)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float a = 0.f;
#pragma unroll
      for (int n = 0; n < 8; ++n) a += sQ[k * 8 + n] * v[n * 8 + j];
      t[k * 8 + j] = a;
    }
#pragma unroll
  for (int k = 0; k < 8; ++k)
#pragma unroll
    for (int l = 0; l < 8; ++l) {
      float a = 0.f;
#pragma unroll
      for (int n = 0; n < 8; ++n) a += t[k * 8 + n] * sQ[l * 8 + n];
      sblk[lb][c][k * 8 + l] = a;
    }
  __syncthreads();
  if (!live) return;
  const float d0 = prm.c_last[c * 3 + 0], d1 = prm.c_last[c * 3 + 1], d2 = prm.c_last[c * 3 + 2];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int yy = by * 8 + i;
    if (yy >= H) break;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int xx = bx * 8 + j;
      if (xx >= W) continue;
      const int p = i * 8 + j;
      y[(((long)b * 3 + c) * H + yy) * W + xx] = d0 * sblk[lb][0][p] + d1 * sblk[lb][1][p] + d2 * sblk[lb][2][p];
    }
  }
}

}  // namespace

extern "C" int aql_jpeg_mask(const float* x, float* y, int B, int H, int W, int keep_y, int keep_u, int keep_v,
                             int backward, hipStream_t stream) {
  AQL_CHECK_ARG(x && y && B > 0 && H > 0 && W > 0, "aql_jpeg_mask: bad args");
  JpegParams prm;
  float T[64], U[64];  // T[k][n] = dct_coeff(n,k,8);  U[k][n] = idct_coeff(n,k,8)   (jpeg_compression.py:44-50)
  const double pi = 3.14159265358979323846;
  for (int k = 0; k < 8; ++k)
    for (int n = 0; n < 8; ++n) {
      T[k * 8 + n] = (float)cos(pi / 8.0 * (n + 0.5) * k);
      U[k * 8 + n] = (float)(((n == 0 ? -0.5 : 0.0) + cos(pi / 8.0 * (k + 0.5) * n)) * sqrt(1.0 / 16.0));
    }
  const float rgb2yuv[9] = {0.299f, 0.587f, 0.114f, -0.14713f, -0.28886f, 0.436f, 0.615f, -0.51499f, -0.10001f};
  const float yuv2rgb[9] = {1.f, 0.f, 1.13983f, 1.f, -0.39465f, -0.58060f, 1.f, 2.03211f, 0.f};
  for (int i = 0; i < 8; ++i)
    for (int j = 0; j < 8; ++j) {
      prm.P[i * 8 + j] = backward ? U[j * 8 + i] : T[i * 8 + j];
      prm.Q[i * 8 + j] = backward ? T[j * 8 + i] : U[i * 8 + j];
    }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      prm.c_first[i * 3 + j] = backward ? yuv2rgb[j * 3 + i] : rgb2yuv[i * 3 + j];
      prm.c_last[i * 3 + j] = backward ? rgb2yuv[j * 3 + i] : yuv2rgb[i * 3 + j];
    }
  // zig-zag order of jpeg_compression.py:34-35: sort by (i+j, -j if (i+j) odd else j); keep the first `count`
  int order[64][2], n = 0;
  for (int s = 0; s < 15; ++s) {
    int js[8], cnt = 0;
    for (int j = 0; j < 8; ++j)
      if (s - j >= 0 && s - j < 8) js[cnt++] = j;
    if (s % 2) {
      for (int q = cnt - 1; q >= 0; --q) { order[n][0] = s - js[q]; order[n][1] = js[q]; ++n; }
    } else {
      for (int q = 0; q < cnt; ++q) { order[n][0] = s - js[q]; order[n][1] = js[q]; ++n; }
    }
  }
  const int keep[3] = {keep_y, keep_u, keep_v};
  for (int c = 0; c < 3; ++c) {
    unsigned long long m = 0;
    for (int q = 0; q < keep[c] && q < 64; ++q) m |= 1ull << (order[q][0] * 8 + order[q][1]);
    prm.mask[c] = m;
  }
  const long nblk = (long)B * ((H + 7) / 8) * ((W + 7) / 8);
  hipLaunchKernelGGL(jpeg_kernel, dim3((unsigned)((nblk + 63) / 64)), dim3(192), 0, stream, x, y, B, H, W, prm);
  AQL_CHECK_LAUNCH("aql_jpeg_mask");
  return AQL_OK;
}
