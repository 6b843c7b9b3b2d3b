// Stage-1 (latent watermark pre-training, train/latent_wm_pretrain.py:159-225) kernels that the PPFT path does not need:
//   * SecretEncoder backward (utils/models.py:51-81 trained at :172, :221)
//   * PRVL_loss forward/backward (latent_wm_pretrain.py:42-50): max over 32x32 windows of the channel-mean |a - b|
// fp32 like the reference.  All tensors here are tiny (B <= 16, 4x64x64 latents) or one pass over a 512x512 image.
#include "aql_common.h"

namespace {

__device__ __forceinline__ float sigmoid_(float z) { return 1.f / (1.f + __expf(-z)); }

// d pre[b,j] for the Linear -> SiLU -> (view R x R, repeat 4 ch, nearest x up) -> conv3x3(4->4) encoder.
// dout [B,4,res,res] already includes the caller's scale.  One thread per (b, j = hidden pixel).
__global__ __launch_bounds__(256) void secenc_dpre_kernel(const float* __restrict__ dout, const float* __restrict__ msg,
                                                          const float* __restrict__ lin_w, const float* __restrict__ lin_b,
                                                          const float* __restrict__ cw, int bits, int R, int res,
                                                          float* __restrict__ dpre) {
  const int b = blockIdx.y;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= R * R) return;
  const int up = res / R;
  const int hy = j / R, hx = j - hy * R;
  float dh = 0.f;
  for (int py = hy * up; py < (hy + 1) * up; ++py)
    for (int px = hx * up; px < (hx + 1) * up; ++px)
      for (int kh = 0; kh < 3; ++kh)
        for (int kw = 0; kw < 3; ++kw) {
          const int oy = py - kh + 1, ox = px - kw + 1;  // output pixel that reads (py,px) through tap (kh,kw)
          if (oy < 0 || ox < 0 || oy >= res || ox >= res) continue;
          for (int co = 0; co < 4; ++co) {
            float wsum = 0.f;
            for (int ci = 0; ci < 4; ++ci) wsum += cw[((co * 4 + ci) * 3 + kh) * 3 + kw];
            dh += wsum * dout[(((long)b * 4 + co) * res + oy) * res + ox];
          }
        }
  float pre = lin_b[j];
  for (int i = 0; i < bits; ++i) pre += lin_w[j * bits + i] * msg[b * bits + i];
  const float s = sigmoid_(pre);
  dpre[b * R * R + j] = dh * s * (1.f + pre * (1.f - s));
}

// dlin_w[j,k] = sum_b dpre[b,j] * msg[b,k];  dlin_b[j] = sum_b dpre[b,j]
__global__ __launch_bounds__(256) void secenc_dlin_kernel(const float* __restrict__ dpre, const float* __restrict__ msg,
                                                          int nb, int bits, int RR, float* __restrict__ dw,
                                                          float* __restrict__ db) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= RR * (bits + 1)) return;
  const int j = id / (bits + 1), k = id - j * (bits + 1);
  float acc = 0.f;
  for (int b = 0; b < nb; ++b) acc += dpre[b * RR + j] * (k < bits ? msg[b * bits + k] : 1.f);
  if (k < bits) dw[j * bits + k] = acc;
  else db[j] = acc;
}

// dconv_w[co,ci,kh,kw] = sum_{b,y,x} dout[b,co,y,x] * u[b,y+kh-1,x+kw-1] (u = upsampled hidden, identical for every ci);
// dconv_b[co] = sum dout.  One workgroup per (co, tap) plus 4 for the bias; deterministic tree reduction.
__global__ __launch_bounds__(256) void secenc_dconv_kernel(const float* __restrict__ dout, const float* __restrict__ hid,
                                                           int nb, int R, int res, float* __restrict__ dcw,
                                                           float* __restrict__ dcb) {
  __shared__ float red[256];
  const int item = blockIdx.x;  // 0..35: (co, kh, kw); 36..39: bias of co = item - 36
  const bool is_bias = item >= 36;
  const int co = is_bias ? item - 36 : item / 9, tap = is_bias ? 0 : item % 9;
  const int kh = tap / 3, kw = tap - kh * 3, up = res / R;
  float acc = 0.f;
  const long n = (long)nb * res * res;
  for (long id = threadIdx.x; id < n; id += 256) {
    const int x = (int)(id % res), y = (int)((id / res) % res);
    const long b = id / ((long)res * res);
    const float g = dout[((b * 4 + co) * res + y) * res + x];
    if (is_bias) {
      acc += g;
    } else {
      const int yy = y + kh - 1, xx = x + kw - 1;
      if (yy < 0 || xx < 0 || yy >= res || xx >= res) continue;
      acc += g * hid[b * R * R + (yy / up) * R + (xx / up)];
    }
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (is_bias) dcb[co] = red[0];
    else
      for (int ci = 0; ci < 4; ++ci) dcw[((co * 4 + ci) * 3 + kh) * 3 + kw] = red[0];
  }
}

// ---- PRVL loss ----------------------------------------------------------------------------------------------------------
// d[b,y,x] = mean_c |a - b|
__global__ __launch_bounds__(256) void prvl_absdiff_kernel(const float* __restrict__ a, const float* __restrict__ b2, int B,
                                                           int C, long HW, float* __restrict__ d) {
  const long n = (long)B * HW;
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < n; id += (long)gridDim.x * blockDim.x) {
    const long bi = id / HW, p = id - bi * HW;
    float acc = 0.f;
    for (int c = 0; c < C; ++c) acc += fabsf(a[(bi * C + c) * HW + p] - b2[(bi * C + c) * HW + p]);
    d[id] = acc / (float)C;
  }
}

// horizontal window sums: rows[b,y,ox] = sum_{x = ox-pad .. ox-pad+win-1} d[b,y,x], ox in [0, Wo)  (zero padding)
__global__ __launch_bounds__(256) void prvl_rows_kernel(const float* __restrict__ d, int B, int H, int W, int win, int pad,
                                                        int Wo, float* __restrict__ rows) {
  const long n = (long)B * H * Wo;
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < n; id += (long)gridDim.x * blockDim.x) {
    const int ox = (int)(id % Wo);
    const long row = id / Wo;
    const float* p = d + row * W;
    float acc = 0.f;
    const int x0 = max(ox - pad, 0), x1 = min(ox - pad + win, W);
    for (int x = x0; x < x1; ++x) acc += p[x];
    rows[id] = acc;
  }
}

// vertical window sums + per-workgroup (max, argmax) of win[b,oy,ox] / win^2
__global__ __launch_bounds__(256) void prvl_cols_max_kernel(const float* __restrict__ rows, int B, int H, int win, int pad,
                                                            int Ho, int Wo, float* __restrict__ bmax,
                                                            long* __restrict__ bidx) {
  __shared__ float sv[256];
  __shared__ long si[256];
  const long n = (long)B * Ho * Wo;
  float best = -1.f;
  long besti = 0;
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < n; id += (long)gridDim.x * blockDim.x) {
    const int ox = (int)(id % Wo), oy = (int)((id / Wo) % Ho);
    const long b = id / ((long)Wo * Ho);
    float acc = 0.f;
    const int y0 = max(oy - pad, 0), y1 = min(oy - pad + win, H);
    for (int y = y0; y < y1; ++y) acc += rows[(b * H + y) * Wo + ox];
    if (acc > best || (acc == best && id < besti)) best = acc, besti = id;
  }
  sv[threadIdx.x] = best;
  si[threadIdx.x] = besti;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      const float o = sv[threadIdx.x + s];
      const long oi = si[threadIdx.x + s];
      if (o > sv[threadIdx.x] || (o == sv[threadIdx.x] && oi < si[threadIdx.x])) sv[threadIdx.x] = o, si[threadIdx.x] = oi;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) bmax[blockIdx.x] = sv[0], bidx[blockIdx.x] = si[0];
}

__global__ void prvl_final_kernel(const float* __restrict__ bmax, const long* __restrict__ bidx, int nblk, float inv_area,
                                  float* __restrict__ loss, long* __restrict__ arg) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float best = -1.f;
  long besti = 0;
  for (int i = 0; i < nblk; ++i)
    if (bmax[i] > best || (bmax[i] == best && bidx[i] < besti)) best = bmax[i], besti = bidx[i];
  loss[0] = best * inv_area;
  arg[0] = besti;
}

// d loss / d a = g * sign(a - b) / (C * win^2) inside the arg-max window, 0 elsewhere (and the negative for b)
__global__ __launch_bounds__(256) void prvl_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b2,
                                                       const long* __restrict__ arg, const float* __restrict__ gout, int B,
                                                       int C, int H, int W, int win, int pad, int Ho, int Wo,
                                                       float* __restrict__ da, float* __restrict__ db) {
  const long n = (long)B * C * H * W;
  const long w = arg[0];
  const int ox = (int)(w % Wo), oy = (int)((w / Wo) % Ho);
  const long wb = w / ((long)Wo * Ho);
  const float g = gout[0] / ((float)C * win * win);
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < n; id += (long)gridDim.x * blockDim.x) {
    const int x = (int)(id % W), y = (int)((id / W) % H);
    const long bi = id / ((long)W * H * C);
    float v = 0.f;
    if (bi == wb && y >= oy - pad && y < oy - pad + win && x >= ox - pad && x < ox - pad + win) {
      const float df = a[id] - b2[id];
      v = df > 0.f ? g : (df < 0.f ? -g : 0.f);
    }
    if (da) da[id] = v;
    if (db) db[id] = -v;
  }
}

inline int grid_for(long n, int cap = 4096) {
  long b = (n + 255) / 256;
  if (b > cap) b = cap;
  return (int)(b < 1 ? 1 : b);
}

}  // namespace

// Gradients of SecretEncoder.encode(): dout [nb,4,res,res] (d loss / d encode output), hidden = the forward's SiLU
// output [nb, R*R]; writes dlin_w [R*R,bits], dlin_b [R*R], dconv_w [4,4,3,3], dconv_b [4]; dpre_scratch nb*R*R floats.
extern "C" int aql_secret_encoder_bwd(const float* dout, const float* msg, const float* lin_w, const float* lin_b,
                                      const float* conv_w, const float* hidden, int nb, int bits, int base_res, int res,
                                      float* dpre_scratch, float* dlin_w, float* dlin_b, float* dconv_w, float* dconv_b,
                                      hipStream_t stream) {
  AQL_CHECK_ARG(dout && msg && lin_w && lin_b && conv_w && hidden && dpre_scratch && dlin_w && dlin_b && dconv_w && dconv_b,
                "aql_secret_encoder_bwd: null");
  AQL_CHECK_ARG(res % base_res == 0 && nb > 0, "aql_secret_encoder_bwd: bad shape");
  const int RR = base_res * base_res;
  hipLaunchKernelGGL(secenc_dpre_kernel, dim3((RR + 255) / 256, nb), dim3(256), 0, stream, dout, msg, lin_w, lin_b, conv_w,
                     bits, base_res, res, dpre_scratch);
  hipLaunchKernelGGL(secenc_dlin_kernel, dim3((RR * (bits + 1) + 255) / 256), dim3(256), 0, stream, dpre_scratch, msg, nb,
                     bits, RR, dlin_w, dlin_b);
  hipLaunchKernelGGL(secenc_dconv_kernel, dim3(40), dim3(256), 0, stream, dout, hidden, nb, base_res, res, dconv_w, dconv_b);
  AQL_CHECK_LAUNCH("aql_secret_encoder_bwd");
  return AQL_OK;
}

extern "C" long aql_prvl_scratch_floats(int B, int H, int W, int win) {
  const int Wo = W + 2 * (win / 2) - win + 1;
  return (long)B * H * W + (long)B * H * Wo + 3 * 1024 + 2;  // absdiff map, row sums, per-workgroup (max, index)
}

// PRVL_loss(img1, img2): loss (device scalar) = max over all (b, window) of the win x win box mean (zero padding win/2)
// of mean_c |img1 - img2|; arg (device long) records the winning window for the backward.
extern "C" int aql_prvl_loss_fwd(const float* img1, const float* img2, int B, int C, int H, int W, int win, float* scratch,
                                 float* loss, long* arg, hipStream_t stream) {
  AQL_CHECK_ARG(img1 && img2 && scratch && loss && arg && B > 0 && C > 0 && win > 0, "aql_prvl_loss_fwd: bad args");
  const int pad = win / 2, Ho = H + 2 * pad - win + 1, Wo = W + 2 * pad - win + 1;
  float* d = scratch;
  float* rows = d + (long)B * H * W;
  float* bmax = scratch + ((((long)B * H * W + (long)B * H * Wo) + 1) & ~1L);  // 8-byte aligned for the index array
  long* bidx = reinterpret_cast<long*>(bmax + 1024);
  const int nblk = grid_for((long)B * Ho * Wo, 1024);
  hipLaunchKernelGGL(prvl_absdiff_kernel, dim3(grid_for((long)B * H * W)), dim3(256), 0, stream, img1, img2, B, C,
                     (long)H * W, d);
  hipLaunchKernelGGL(prvl_rows_kernel, dim3(grid_for((long)B * H * Wo)), dim3(256), 0, stream, d, B, H, W, win, pad, Wo,
                     rows);
  hipLaunchKernelGGL(prvl_cols_max_kernel, dim3(nblk), dim3(256), 0, stream, rows, B, H, win, pad, Ho, Wo, bmax, bidx);
  hipLaunchKernelGGL(prvl_final_kernel, dim3(1), dim3(64), 0, stream, bmax, bidx, nblk, 1.f / ((float)win * win), loss, arg);
  AQL_CHECK_LAUNCH("aql_prvl_loss_fwd");
  return AQL_OK;
}

// d1 / d2 (either may be null) = gout * d loss / d img1, d img2
extern "C" int aql_prvl_loss_bwd(const float* img1, const float* img2, const long* arg, const float* gout, int B, int C,
                                 int H, int W, int win, float* d1, float* d2, hipStream_t stream) {
  AQL_CHECK_ARG(img1 && img2 && arg && gout && (d1 || d2), "aql_prvl_loss_bwd: bad args");
  const int pad = win / 2, Ho = H + 2 * pad - win + 1, Wo = W + 2 * pad - win + 1;
  hipLaunchKernelGGL(prvl_bwd_kernel, dim3(grid_for((long)B * C * H * W)), dim3(256), 0, stream, img1, img2, arg, gout, B,
                     C, H, W, win, pad, Ho, Wo, d1, d2);
  AQL_CHECK_LAUNCH("aql_prvl_loss_bwd");
  return AQL_OK;
}
