// HBM-bound elementwise / reduction kernels of the PPFT step: GEGLU, nearest-x2 upsample backward,
// forward-diffusion noising, MSE loss, MapperNet, SecretEncoder, LoRA weight casts, gradient-norm clip + AdamW.
#include "aql_common.h"

namespace {

__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
  f[0] = bf16lo(v.x); f[1] = bf16hi(v.x); f[2] = bf16lo(v.y); f[3] = bf16hi(v.y);
  f[4] = bf16lo(v.z); f[5] = bf16hi(v.z); f[6] = bf16lo(v.w); f[7] = bf16hi(v.w);
}
__device__ __forceinline__ uint4 pack8(const float* f) {
  return make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7]));
}

__device__ __forceinline__ float block_sum(float v, float* sm) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  if ((threadIdx.x & 63) == 0) sm[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += sm[i];
  __syncthreads();
  return t;
}

// ---- GEGLU (scripts/lib/original_unet.py:727-729): out = h * gelu(g), exact erf gelu ----------------
__global__ __launch_bounds__(256) void geglu_fwd_kernel(const bf16_t* __restrict__ in, long M, int F,
                                                        bf16_t* __restrict__ out) {
  const int cols = F >> 3;
  const long n = M * cols;
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < n; id += (long)gridDim.x * blockDim.x) {
    const long m = id / cols;
    const int c = (int)(id - m * cols) * 8;
    float h[8], g[8], o[8];
    unpack8(*reinterpret_cast<const uint4*>(in + m * 2 * F + c), h);
    unpack8(*reinterpret_cast<const uint4*>(in + m * 2 * F + F + c), g);
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
      const aql_f32x2_t a = aql_f32x2_t{h[j], h[j + 1]} * aql_gelu2(aql_f32x2_t{g[j], g[j + 1]});
      o[j] = a.x;
      o[j + 1] = a.y;
    }
    *reinterpret_cast<uint4*>(out + m * F + c) = pack8(o);
  }
}

__global__ __launch_bounds__(256) void geglu_bwd_kernel(const bf16_t* __restrict__ in, const bf16_t* __restrict__ dy,
                                                        long M, int F, bf16_t* __restrict__ din) {
  const int cols = F >> 3;
  const long n = M * cols;
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < n; id += (long)gridDim.x * blockDim.x) {
    const long m = id / cols;
    const int c = (int)(id - m * cols) * 8;
    float h[8], g[8], d[8], dh[8], dg[8];
    unpack8(*reinterpret_cast<const uint4*>(in + m * 2 * F + c), h);
    unpack8(*reinterpret_cast<const uint4*>(in + m * 2 * F + F + c), g);
    unpack8(*reinterpret_cast<const uint4*>(dy + m * F + c), d);
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
      aql_f32x2_t a, b;
      aql_geglu_bwd2(aql_f32x2_t{d[j], d[j + 1]}, aql_f32x2_t{h[j], h[j + 1]}, aql_f32x2_t{g[j], g[j + 1]}, a, b);
      dh[j] = a.x, dh[j + 1] = a.y;
      dg[j] = b.x, dg[j + 1] = b.y;
    }
    *reinterpret_cast<uint4*>(din + m * 2 * F + c) = pack8(dh);
    *reinterpret_cast<uint4*>(din + m * 2 * F + F + c) = pack8(dg);
  }
}

// ---- backward of the nearest x2 upsample folded into conv3x3: dX[b,h,w,:] = sum of the 2x2 block ------
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const bf16_t* __restrict__ du, int B, int H, int W, int C,
                                                             bf16_t* __restrict__ dx) {
  const int cols = C >> 3;
  const long n = (long)B * H * W * cols;
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < n; id += (long)gridDim.x * blockDim.x) {
    const int c = (int)(id % cols) * 8;
    long p = id / cols;
    const int w = (int)(p % W);
    p /= W;
    const int h = (int)(p % H);
    const int b = (int)(p / H);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float v[8];
        unpack8(*reinterpret_cast<const uint4*>(du + (((long)b * 2 * H + 2 * h + i) * 2 * W + 2 * w + j) * C + c), v);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += v[e];
      }
    *reinterpret_cast<uint4*>(dx + (((long)b * H + h) * W + w) * C + c) = pack8(acc);
  }
}

// ---- skip-connection concat on channels-last maps (original_unet.py:1133,1224): out[p][0:Ca] = a[p], out[p][Ca:] = b[p] in ONE
// launch, and its backward (one launch, two dense outputs).  DIR 0: (a, b) -> cat;  DIR 1: cat -> (a, b).  Four 16-byte chunks
// per thread in flight.
template <int DIR>
__global__ __launch_bounds__(256) void cat_channels_kernel(bf16_t* __restrict__ a, bf16_t* __restrict__ b, bf16_t* __restrict__ cat,
                                                           long npix, int Ca, int Cb) {
  const int ca = Ca >> 3, cols = (Ca + Cb) >> 3;
  const long n = npix * cols, stride = (long)gridDim.x * blockDim.x;
  for (long id0 = (long)blockIdx.x * blockDim.x + threadIdx.x; id0 < n; id0 += 4 * stride) {
    uint4 v[4];
    bf16_t* part[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long id = id0 + u * stride < n ? id0 + u * stride : id0;
      const long p = id / cols;
      const int c = (int)(id - p * cols);
      part[u] = c < ca ? a + p * Ca + c * 8 : b + p * Cb + (c - ca) * 8;
      v[u] = *reinterpret_cast<const uint4*>(DIR == 0 ? part[u] : cat + id * 8);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long id = id0 + u * stride;
      if (id >= n) continue;
      *reinterpret_cast<uint4*>(DIR == 0 ? cat + id * 8 : part[u]) = v[u];
    }
  }
}

// ---- add_noise (diffusers DDPMScheduler.add_noise via utils/cschedulers.py:15; ppft_train.py:1010-1011) -----
// out = sqrt(acp[t]) * x + sqrt(1-acp[t]) * eps, computed in fp32, stored bf16.  Two inputs share eps and t.
__global__ __launch_bounds__(256) void add_noise_kernel(const float* __restrict__ x0, const float* __restrict__ wm,
                                                        const float* __restrict__ eps, const long* __restrict__ t,
                                                        const float* __restrict__ acp, int per_sample,
                                                        bf16_t* __restrict__ noisy, bf16_t* __restrict__ noisy_wm,
                                                        long n) {
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < n; id += (long)gridDim.x * blockDim.x) {
    const int b = (int)(id / per_sample);
    const float a = acp[t[b]];
    const float sa = sqrtf(a), sb = sqrtf(1.f - a);
    const float e = eps[id];
    noisy[id] = f32_to_bf16(sa * x0[id] + sb * e);
    if (noisy_wm != nullptr) noisy_wm[id] = f32_to_bf16(sa * (x0[id] + wm[id]) + sb * e);
  }
}

// ---- PPFT step prologue in ONE launch (train/ppft_train.py:994-1011 + the head of the U-Net forward, original_unet.py:323-361)
// Everything between the batch and the first GEMM of the twin (clean | watermarked) forward is a handful of tiny independent
// element-wise jobs: 15 launches of 4-6 us each on the critical path of the step (profiles/r03: 0.4 ms from the start of the graph to
// the first U-Net kernel).  Jobs, by block range:
//   noise : x2 [2B][H][W][8] bf16 (channels-last, 4 latent channels + 4 zero channels = conv_in's packed width):
//           first half  sqrt(acp[t]) z        + sqrt(1 - acp[t]) eps      (the clean pass sees the un-watermarked latents)
//           second half sqrt(acp[t]) (z + wm) + sqrt(1 - acp[t]) eps      -- aql_add_noise's arithmetic, both roundings identical
//   ctx   : ctx2 [2B][L][D] bf16 = the text states twice (fp32 or bf16 in)
//   temb  : temb [2B][2 half] bf16 = [cos(t f_i) | sin(t f_i)], timesteps repeated for the second half; f from the caller's table
//           (get_timestep_embedding with flip_sin_to_cos: fp32 product, cosf / sinf, one rounding to bf16)
//   map   : S32 [B][r] = sum_i msg_i E[i,:] / sqrt(bits) + 1 (MapperNet, utils/models.py:110-115); S16 [2B][r] bf16 = [0 | S32];
//           ds [B][r] fp32 = 0 (the step's dS accumulator)
struct PrologueArgs {
  const float *z, *wm, *eps, *acp, *msg, *E, *freq;
  const long* t;
  const void* ctx;
  bf16_t *x2, *ctx2, *temb, *S16;
  float *S32, *ds;
  int B, HW, bits, r, half, ctx_elems, ctx_f32;
  int nb_noise, nb_ctx, nb_temb;
};
__global__ __launch_bounds__(256) void ppft_prologue_kernel(const PrologueArgs a) {
  int blk = blockIdx.x;
  const int tid = threadIdx.x;
  if (blk < a.nb_noise) {   // one thread per (sample, pixel): 4 strided fp32 reads per operand, one 16-byte store per half
    const int id = blk * 256 + tid;
    if (id >= a.B * a.HW) return;
    const int b = id / a.HW, p = id - b * a.HW;
    const float ac = a.acp[a.t[b]];
    const float sa = sqrtf(ac), sb = sqrtf(1.f - ac);
    float xc[4], xw[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const long src = ((long)b * 4 + c) * a.HW + p;
      const float x0 = a.z[src], e = a.eps[src];
      xc[c] = sa * x0 + sb * e;
      xw[c] = sa * (x0 + a.wm[src]) + sb * e;
    }
    const uint4 vc = make_uint4(pack_bf16x2(xc[0], xc[1]), pack_bf16x2(xc[2], xc[3]), 0u, 0u);
    const uint4 vw = make_uint4(pack_bf16x2(xw[0], xw[1]), pack_bf16x2(xw[2], xw[3]), 0u, 0u);
    *reinterpret_cast<uint4*>(a.x2 + (long)id * 8) = vc;
    *reinterpret_cast<uint4*>(a.x2 + ((long)a.B * a.HW + id) * 8) = vw;
    return;
  }
  blk -= a.nb_noise;
  if (blk < a.nb_ctx) {   // 8 elements per thread
    const long id = ((long)blk * 256 + tid) * 8;
    if (id >= a.ctx_elems) return;
    uint4 v;
    if (a.ctx_f32) {
      const float4 lo = *reinterpret_cast<const float4*>(static_cast<const float*>(a.ctx) + id);
      const float4 hi = *reinterpret_cast<const float4*>(static_cast<const float*>(a.ctx) + id + 4);
      v = make_uint4(pack_bf16x2(lo.x, lo.y), pack_bf16x2(lo.z, lo.w), pack_bf16x2(hi.x, hi.y), pack_bf16x2(hi.z, hi.w));
    } else {
      v = *reinterpret_cast<const uint4*>(static_cast<const bf16_t*>(a.ctx) + id);
    }
    *reinterpret_cast<uint4*>(a.ctx2 + id) = v;
    *reinterpret_cast<uint4*>(a.ctx2 + a.ctx_elems + id) = v;
    return;
  }
  blk -= a.nb_ctx;
  if (blk < a.nb_temb) {
    const int id = blk * 256 + tid;
    if (id >= 2 * a.B * a.half) return;
    const int row = id / a.half, i = id - row * a.half;
    const float arg = (float)a.t[row % a.B] * a.freq[i];
    a.temb[(long)row * 2 * a.half + i] = f32_to_bf16(cosf(arg));
    a.temb[(long)row * 2 * a.half + a.half + i] = f32_to_bf16(sinf(arg));
    return;
  }
  // mapper: one thread per (sample, rank column); the bits-long dot product with all loads in flight
  for (int id = tid; id < a.B * a.r; id += 256) {
    const int b = id / a.r, j = id - b * a.r;
    float acc = 0.f;
    for (int i0 = 0; i0 < a.bits; i0 += 16) {
      float e[16], m[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int i = min(i0 + u, a.bits - 1);
        e[u] = a.E[i * a.r + j];
        m[u] = a.msg[b * a.bits + i];
      }
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (i0 + u < a.bits) acc += e[u] * m[u];   // same order as mapper_fwd_kernel: bit-identical S
    }
    const float v = acc * rsqrtf((float)a.bits) + 1.f;
    a.S32[id] = v;
    a.S16[id] = 0;
    a.S16[a.B * a.r + id] = f32_to_bf16(v);
    a.ds[id] = 0.f;
  }
}

// ---- MSE loss (ppft_train.py:1051): loss = mean((p-t)^2) in fp32; dpred = 2 (p-t)/n ------------------
__global__ __launch_bounds__(256) void mse_kernel(const bf16_t* __restrict__ p, const bf16_t* __restrict__ t, long n,
                                                  float* __restrict__ loss, bf16_t* __restrict__ dpred) {
  __shared__ float sm[4];
  float acc = 0.f;
  const float k = 2.f / (float)n;
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < n; id += (long)gridDim.x * blockDim.x) {
    const float d = bf16_to_f32(p[id]) - bf16_to_f32(t[id]);
    acc += d * d;
    if (dpred != nullptr) dpred[id] = f32_to_bf16(k * d);
  }
  const float s = block_sum(acc, sm);
  if (threadIdx.x == 0) atomicAdd(loss, s / (float)n);
}

// ---- MapperNet (utils/models.py:110-115): S = sum_i m_i E[i,:] / sqrt(bits) + 1 ---------------------
__global__ void mapper_fwd_kernel(const float* __restrict__ msg, const float* __restrict__ E, int bits, int r,
                                  float* __restrict__ S32, bf16_t* __restrict__ S16) {
  const int b = blockIdx.x;
  const float inv = rsqrtf((float)bits);
  for (int j = threadIdx.x; j < r; j += blockDim.x) {
    float acc = 0.f;
    for (int i = 0; i < bits; ++i) acc += E[i * r + j] * msg[b * bits + i];
    const float v = acc * inv + 1.f;
    if (S32 != nullptr) S32[b * r + j] = v;
    if (S16 != nullptr) S16[b * r + j] = f32_to_bf16(v);
  }
}
// dE[i,j] += sum_b dS[b,j] * m[b,i] / sqrt(bits)
__global__ void mapper_bwd_kernel(const float* __restrict__ msg, const float* __restrict__ dS, int nb, int bits, int r,
                                  float* __restrict__ dE) {
  const int i = blockIdx.x;
  const float inv = rsqrtf((float)bits);
  for (int j = threadIdx.x; j < r; j += blockDim.x) {
    float acc = 0.f;
    for (int b = 0; b < nb; ++b) acc += dS[b * r + j] * msg[b * bits + i];
    dE[i * r + j] += acc * inv;
  }
}

// ---- SecretEncoder (utils/models.py:57-64,74-81): Linear(bits->R*R) -> SiLU -> [1,R,R] repeated to 4 ch
//      -> nearest x(res/R) -> conv3x3(4->4, pad 1).  (The bilinear resize to the latent size is the identity
//      when the latent is res x res.)  Output NCHW fp32 [B,4,res,res] times `out_scale`.
__global__ void secret_hidden_kernel(const float* __restrict__ msg, const float* __restrict__ W,
                                     const float* __restrict__ bvec, int bits, int RR, float* __restrict__ hid) {
  const int b = blockIdx.y;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= RR) return;
  float acc = bvec[j];
  for (int i = 0; i < bits; ++i) acc += W[j * bits + i] * msg[b * bits + i];
  hid[b * RR + j] = acc / (1.f + __expf(-acc));
}
__global__ void secret_conv_kernel(const float* __restrict__ hid, const float* __restrict__ cw,
                                   const float* __restrict__ cb, int R, int res, float out_scale,
                                   float* __restrict__ out) {
  const int b = blockIdx.y;
  const int up = res / R;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 4 * res * res) return;
  const int co = idx / (res * res);
  const int y = (idx / res) % res, x = idx % res;
  float acc = cb[co];
  for (int kh = 0; kh < 3; ++kh)
    for (int kw = 0; kw < 3; ++kw) {
      const int yy = y + kh - 1, xx = x + kw - 1;
      if (yy < 0 || xx < 0 || yy >= res || xx >= res) continue;
      const float v = hid[b * R * R + (yy / up) * R + (xx / up)];  // same value in all 4 input channels
      float wsum = 0.f;
      for (int ci = 0; ci < 4; ++ci) wsum += cw[((co * 4 + ci) * 3 + kh) * 3 + kw];
      acc += v * wsum;
    }
  out[(long)b * 4 * res * res + idx] = acc * out_scale;
}

// ---- LoRA master weights (fp32, [rows, cols]) -> bf16 copy and bf16 transposed copy ------------------
__global__ __launch_bounds__(256) void cast_transpose_kernel(const float* __restrict__ w, int rows, int cols,
                                                             bf16_t* __restrict__ out, bf16_t* __restrict__ outT) {
  __shared__ float tile[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int i = ty; i < 32; i += 8) {
    const int r = by + i, c = bx + tx;
    float v = 0.f;
    if (r < rows && c < cols) {
      v = w[(long)r * cols + c];
      out[(long)r * cols + c] = f32_to_bf16(v);
    }
    tile[i][tx] = v;
  }
  __syncthreads();
  if (outT != nullptr) {
    for (int i = ty; i < 32; i += 8) {
      const int c = bx + i, r = by + tx;
      if (r < rows && c < cols) outT[(long)c * rows + r] = f32_to_bf16(tile[tx][i]);
    }
  }
}

// batched form: one launch for every LoRA tensor.  desc[i] = {src, out, outT, rows, cols, first_tile}; 32x32 tiles.
struct CastDesc {
  const float* w;
  bf16_t* out;
  bf16_t* outT;
  int rows, cols;
  int first_tile, pad;   // pad: leading dimension of outT, 0 = rows (dense)
};
__global__ __launch_bounds__(256) void cast_transpose_batched_kernel(const CastDesc* __restrict__ desc, int n) {
  __shared__ float tile[32][33];
  int lo = 0, hi = n - 1;
  while (lo < hi) {  // last descriptor with first_tile <= blockIdx.x
    const int mid = (lo + hi + 1) >> 1;
    if (desc[mid].first_tile <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const CastDesc d = desc[lo];
  const int t = blockIdx.x - d.first_tile;
  const int tx_tiles = (d.cols + 31) / 32;
  const int bx = (t % tx_tiles) * 32, by = (t / tx_tiles) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const long ldt = d.pad > 0 ? d.pad : d.rows;   // leading dimension of the transposed copy (column blocks of a wider matrix: LoraBank)
  for (int i = ty; i < 32; i += 8) {
    const int r = by + i, c = bx + tx;
    float v = 0.f;
    if (r < d.rows && c < d.cols) {
      v = d.w[(long)r * d.cols + c];
      d.out[(long)r * d.cols + c] = f32_to_bf16(v);
    }
    tile[i][tx] = v;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = bx + i, r = by + tx;
    if (r < d.rows && c < d.cols) d.outT[(long)c * ldt + r] = f32_to_bf16(tile[tx][i]);
  }
}

// ---- dS[b, j] = sum_{m in sample b} dTs[m, j] * T[m, j]  (gradient of the per-message diagonal) --------
__global__ __launch_bounds__(256) void lora_ds_kernel(const bf16_t* __restrict__ dTs, const bf16_t* __restrict__ T,
                                                      int rows_per_sample, int r, float* __restrict__ dS) {
  // grid: (slabs, nb); block 256 threads = (r/8 chunk columns) x row lanes; atomics into dS
  __shared__ float acc[1024];
  const int b = blockIdx.y;
  const int cols = r >> 3;
  const int rp = blockDim.x / cols;
  const int col = threadIdx.x % cols, rr = threadIdx.x / cols;
  const int slab_rows = (rows_per_sample + gridDim.x - 1) / gridDim.x;
  const int r0 = blockIdx.x * slab_rows, r1 = min(rows_per_sample, r0 + slab_rows);
  float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (rr < rp) {
    for (int m = r0 + rr; m < r1; m += rp) {
      const long off = ((long)b * rows_per_sample + m) * r + col * 8;
      float x[8], y[8];
      unpack8(*reinterpret_cast<const uint4*>(dTs + off), x);
      unpack8(*reinterpret_cast<const uint4*>(T + off), y);
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] += x[j] * y[j];
    }
  }
  for (int i = threadIdx.x; i < r; i += blockDim.x) acc[i] = 0.f;
  __syncthreads();
  if (rr < rp) {
#pragma unroll
    for (int j = 0; j < 8; ++j) atomicAdd(&acc[col * 8 + j], a[j]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < r; i += blockDim.x) atomicAdd(&dS[b * r + i], acc[i]);
}

// grouped form: all LoRA sites of a backward pass in one launch
struct DsGroupDesc {
  const bf16_t* dTs;
  const bf16_t* T;
  float* dS;
  int nb, rps, r, slabs;
  int first_block, pad;   // pad: row stride of dTs in elements, 0 = r (dense) -- a column block of a stacked [M, G r] product (round 6)
};
__global__ __launch_bounds__(256) void lora_ds_grouped_kernel(const DsGroupDesc* __restrict__ descs, int n) {
  __shared__ float acc[1024];
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (descs[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const DsGroupDesc d = descs[lo];
  const int local = blockIdx.x - d.first_block;
  const int slab = local % d.slabs, b = local / d.slabs;
  const int r = d.r, cols = r >> 3;
  const int rp = 256 / cols;
  const int col = threadIdx.x % cols, rr = threadIdx.x / cols;
  const int slab_rows = (d.rps + d.slabs - 1) / d.slabs;
  const int r0 = slab * slab_rows, r1 = min(d.rps, r0 + slab_rows);
  float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (rr < rp) {
    for (int m = r0 + rr; m < r1; m += rp) {
      const long row = (long)b * d.rps + m, off = row * r + col * 8;
      float x[8], y[8];
      unpack8(*reinterpret_cast<const uint4*>(d.dTs + (d.pad > 0 ? row * d.pad + col * 8 : off)), x);
      unpack8(*reinterpret_cast<const uint4*>(d.T + off), y);
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] += x[j] * y[j];
    }
  }
  for (int i = threadIdx.x; i < r; i += blockDim.x) acc[i] = 0.f;
  __syncthreads();
  if (rr < rp) {
#pragma unroll
    for (int j = 0; j < 8; ++j) atomicAdd(&acc[col * 8 + j], a[j]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < r; i += blockDim.x) atomicAdd(&d.dS[b * r + i], acc[i]);
}

// ---- global grad-norm clip + AdamW on flat fp32 buffers (ppft_train.py:1059-1068, 779-787) -----------
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long n, float* __restrict__ out) {
  __shared__ float sm[4];
  float acc = 0.f;
  // 16-byte loads, four in flight per thread (scalar 4-byte loads: 34 us for the 54 MB of the rank-32 gradients, 1.6 TB/s)
  const long n4 = (reinterpret_cast<uintptr_t>(g) & 15u) == 0 ? n >> 2 : 0;
  const float4* g4 = reinterpret_cast<const float4*>(g);
  const long stride = (long)gridDim.x * blockDim.x;
  long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; id + 3 * stride < n4; id += 4 * stride) {
    const float4 a = g4[id], b = g4[id + stride], c = g4[id + 2 * stride], d = g4[id + 3 * stride];
    acc += (a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w) + (b.x * b.x + b.y * b.y + b.z * b.z + b.w * b.w) +
           (c.x * c.x + c.y * c.y + c.z * c.z + c.w * c.w) + (d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w);
  }
  for (; id < n4; id += stride) {
    const float4 a = g4[id];
    acc += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
  }
  for (long t = n4 * 4 + (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride) acc += g[t] * g[t];
  const float s = block_sum(acc, sm);
  if (threadIdx.x == 0) atomicAdd(out, s);
}

// torch.nn.utils.clip_grad_norm_: coef = min(1, max_norm / (norm + 1e-6)); torch.optim.AdamW (decoupled decay).
// `sumsq` may be null (no clipping, e.g. the MapperNet group).  step is the 1-based step count.
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, long n,
                                                    const float* __restrict__ sumsq, float max_norm,
                                                    const float* __restrict__ lr_ptr, float beta1, float beta2,
                                                    float eps, float wd, const int* __restrict__ step_ptr) {
  float coef = 1.f;
  if (sumsq != nullptr) {
    const float norm = sqrtf(*sumsq);
    coef = fminf(1.f, max_norm / (norm + 1e-6f));
  }
  const float lr = *lr_ptr;
  const float step = (float)(*step_ptr);
  const float bc1 = 1.f - powf(beta1, step), bc2 = 1.f - powf(beta2, step);
  const float step_size = lr / bc1;
  const float rbc2 = rsqrtf(bc2);
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < n; id += (long)gridDim.x * blockDim.x) {
    const float gr = g[id] * coef;
    float w = p[id];
    w *= (1.f - lr * wd);
    const float mm = beta1 * m[id] + (1.f - beta1) * gr;
    const float vv = beta2 * v[id] + (1.f - beta2) * gr * gr;
    m[id] = mm;
    v[id] = vv;
    const float denom = sqrtf(vv) * rbc2 + eps;
    p[id] = w - step_size * (mm / denom);
  }
}

inline int grid_for(long n, int per = 256, int cap = 2048) {
  long b = (n + per - 1) / per;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

extern "C" int aql_geglu_fwd(const bf16_t* in, long M, int F, bf16_t* out, hipStream_t stream) {
  AQL_CHECK_ARG(in && out && F % 8 == 0, "aql_geglu_fwd: bad args");
  hipLaunchKernelGGL(geglu_fwd_kernel, dim3(grid_for(M * (F / 8))), dim3(256), 0, stream, in, M, F, out);
  AQL_CHECK_LAUNCH("aql_geglu_fwd");
  return AQL_OK;
}
extern "C" int aql_geglu_bwd(const bf16_t* in, const bf16_t* dy, long M, int F, bf16_t* din, hipStream_t stream) {
  AQL_CHECK_ARG(in && dy && din && F % 8 == 0, "aql_geglu_bwd: bad args");
  hipLaunchKernelGGL(geglu_bwd_kernel, dim3(grid_for(M * (F / 8))), dim3(256), 0, stream, in, dy, M, F, din);
  AQL_CHECK_LAUNCH("aql_geglu_bwd");
  return AQL_OK;
}
extern "C" int aql_upsample2x_bwd(const bf16_t* du, int B, int H, int W, int C, bf16_t* dx, hipStream_t stream) {
  AQL_CHECK_ARG(du && dx && C % 8 == 0, "aql_upsample2x_bwd: bad args");
  hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3(grid_for((long)B * H * W * (C / 8))), dim3(256), 0, stream, du, B, H,
                     W, C, dx);
  AQL_CHECK_LAUNCH("aql_upsample2x_bwd");
  return AQL_OK;
}
extern "C" int aql_cat_channels(const bf16_t* a, const bf16_t* b, long npix, int Ca, int Cb, bf16_t* cat, hipStream_t stream) {
  AQL_CHECK_ARG(a && b && cat && npix > 0 && Ca > 0 && Cb > 0 && Ca % 8 == 0 && Cb % 8 == 0, "aql_cat_channels: bad args");
  hipLaunchKernelGGL(cat_channels_kernel<0>, dim3(grid_for((npix * ((Ca + Cb) / 8) + 3) / 4)), dim3(256), 0, stream,
                     const_cast<bf16_t*>(a), const_cast<bf16_t*>(b), cat, npix, Ca, Cb);
  AQL_CHECK_LAUNCH("aql_cat_channels");
  return AQL_OK;
}
extern "C" int aql_split_channels(const bf16_t* cat, long npix, int Ca, int Cb, bf16_t* a, bf16_t* b, hipStream_t stream) {
  AQL_CHECK_ARG(a && b && cat && npix > 0 && Ca > 0 && Cb > 0 && Ca % 8 == 0 && Cb % 8 == 0, "aql_split_channels: bad args");
  hipLaunchKernelGGL(cat_channels_kernel<1>, dim3(grid_for((npix * ((Ca + Cb) / 8) + 3) / 4)), dim3(256), 0, stream, a, b,
                     const_cast<bf16_t*>(cat), npix, Ca, Cb);
  AQL_CHECK_LAUNCH("aql_split_channels");
  return AQL_OK;
}
extern "C" int aql_add_noise(const float* x0, const float* wm, const float* eps, const long* t, const float* acp,
                             int B, int per_sample, bf16_t* noisy, bf16_t* noisy_wm, hipStream_t stream) {
  AQL_CHECK_ARG(x0 && eps && t && acp && noisy && (wm != nullptr || noisy_wm == nullptr), "aql_add_noise: bad args");
  const long n = (long)B * per_sample;
  hipLaunchKernelGGL(add_noise_kernel, dim3(grid_for(n)), dim3(256), 0, stream, x0, wm, eps, t, acp, per_sample, noisy,
                     noisy_wm, n);
  AQL_CHECK_LAUNCH("aql_add_noise");
  return AQL_OK;
}
// See ppft_prologue_kernel.  z / wm / eps [B][4][H*W] fp32 (NCHW), t [B] int64, acp [T] fp32, msg [B][bits] fp32, E [bits][r] fp32,
// freq [half] fp32, ctx [B][ctx_per_sample] (fp32 when ctx_f32 else bf16; ctx_per_sample % 8 == 0).
// Out: x2 [2B][H*W][8] bf16, ctx2 [2B][ctx_per_sample] bf16, temb [2B][2*half] bf16, S32 [B][r], S16 [2B][r] bf16, ds [B][r] = 0.
extern "C" int aql_ppft_prologue(const float* z, const float* wm, const float* eps, const long* t, const float* acp,
                                 const float* msg, const float* E, const float* freq, const void* ctx, int ctx_f32, int B,
                                 int HW, int bits, int r, int half, long ctx_per_sample, bf16_t* x2, bf16_t* ctx2,
                                 bf16_t* temb, float* S32, bf16_t* S16, float* ds, hipStream_t stream) {
  AQL_CHECK_ARG(z && wm && eps && t && acp && msg && E && freq && ctx && x2 && ctx2 && temb && S32 && S16 && ds,
                "aql_ppft_prologue: null operand");
  AQL_CHECK_ARG(B > 0 && HW > 0 && bits > 0 && r > 0 && half > 0 && ctx_per_sample > 0 && ctx_per_sample % 8 == 0 &&
                    (long)B * ctx_per_sample < (1L << 31), "aql_ppft_prologue: bad shape");
  PrologueArgs a{};
  a.z = z, a.wm = wm, a.eps = eps, a.acp = acp, a.msg = msg, a.E = E, a.freq = freq, a.t = t, a.ctx = ctx;
  a.x2 = x2, a.ctx2 = ctx2, a.temb = temb, a.S16 = S16, a.S32 = S32, a.ds = ds;
  a.B = B, a.HW = HW, a.bits = bits, a.r = r, a.half = half, a.ctx_elems = (int)(B * ctx_per_sample), a.ctx_f32 = ctx_f32;
  a.nb_noise = (B * HW + 255) / 256;
  a.nb_ctx = (a.ctx_elems / 8 + 255) / 256;
  a.nb_temb = (2 * B * half + 255) / 256;
  hipLaunchKernelGGL(ppft_prologue_kernel, dim3(a.nb_noise + a.nb_ctx + a.nb_temb + 1), dim3(256), 0, stream, a);
  AQL_CHECK_LAUNCH("aql_ppft_prologue");
  return AQL_OK;
}
extern "C" int aql_mse_fwd_bwd(const bf16_t* pred, const bf16_t* target, long n, float* loss, bf16_t* dpred,
                               hipStream_t stream) {
  AQL_CHECK_ARG(pred && target && loss, "aql_mse_fwd_bwd: bad args");
  (void)hipMemsetAsync(loss, 0, sizeof(float), stream);
  hipLaunchKernelGGL(mse_kernel, dim3(grid_for(n, 256, 256)), dim3(256), 0, stream, pred, target, n, loss, dpred);
  AQL_CHECK_LAUNCH("aql_mse_fwd_bwd");
  return AQL_OK;
}
extern "C" int aql_mapper_fwd(const float* msg, const float* E, int nb, int bits, int r, float* S32, bf16_t* S16,
                              hipStream_t stream) {
  AQL_CHECK_ARG(msg && E && (S32 || S16), "aql_mapper_fwd: bad args");
  hipLaunchKernelGGL(mapper_fwd_kernel, dim3(nb), dim3(256), 0, stream, msg, E, bits, r, S32, S16);
  AQL_CHECK_LAUNCH("aql_mapper_fwd");
  return AQL_OK;
}
extern "C" int aql_mapper_bwd(const float* msg, const float* dS, int nb, int bits, int r, float* dE,
                              hipStream_t stream) {
  AQL_CHECK_ARG(msg && dS && dE, "aql_mapper_bwd: bad args");
  hipLaunchKernelGGL(mapper_bwd_kernel, dim3(bits), dim3(256), 0, stream, msg, dS, nb, bits, r, dE);
  AQL_CHECK_LAUNCH("aql_mapper_bwd");
  return AQL_OK;
}
extern "C" int aql_secret_encoder_fwd(const float* msg, const float* lin_w, const float* lin_b, const float* conv_w,
                                      const float* conv_b, int nb, int bits, int base_res, int res, float out_scale,
                                      float* hidden_scratch, float* out, hipStream_t stream) {
  AQL_CHECK_ARG(msg && lin_w && lin_b && conv_w && conv_b && hidden_scratch && out, "aql_secret_encoder_fwd: null");
  AQL_CHECK_ARG(res % base_res == 0, "aql_secret_encoder_fwd: res must be a multiple of base_res");
  const int RR = base_res * base_res;
  hipLaunchKernelGGL(secret_hidden_kernel, dim3((RR + 255) / 256, nb), dim3(256), 0, stream, msg, lin_w, lin_b, bits, RR,
                     hidden_scratch);
  hipLaunchKernelGGL(secret_conv_kernel, dim3((4 * res * res + 255) / 256, nb), dim3(256), 0, stream, hidden_scratch,
                     conv_w, conv_b, base_res, res, out_scale, out);
  AQL_CHECK_LAUNCH("aql_secret_encoder_fwd");
  return AQL_OK;
}
extern "C" int aql_cast_transpose(const float* w, int rows, int cols, bf16_t* out, bf16_t* outT, hipStream_t stream) {
  AQL_CHECK_ARG(w && out, "aql_cast_transpose: bad args");
  hipLaunchKernelGGL(cast_transpose_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0, stream, w, rows,
                     cols, out, outT);
  AQL_CHECK_LAUNCH("aql_cast_transpose");
  return AQL_OK;
}
extern "C" int aql_cast_transpose_batched(const void* desc, int n, int total_tiles, hipStream_t stream) {
  AQL_CHECK_ARG(desc && n > 0 && total_tiles > 0, "aql_cast_transpose_batched: bad args");
  hipLaunchKernelGGL(cast_transpose_batched_kernel, dim3(total_tiles), dim3(256), 0, stream,
                     static_cast<const CastDesc*>(desc), n);
  AQL_CHECK_LAUNCH("aql_cast_transpose_batched");
  return AQL_OK;
}
// Weight-side form of the watermark-LoRA linear (ops.wside_backward; the reference's branch is utils/lora_modules.py:13-19): after
// P[b][n][j] = sum_k dWe_b[n][k] A[j][k] the two r-wide gradients that remain are reductions over the samples / the output rows:
//     dBup[n][j] += sum_b P[b][n][j] S[b][j]          dS[b][j] += sum_n Bup[n][j] P[b][n][j]
// One workgroup per 16 rows n, one thread per rank column j (coalesced along j); dBup is owned (no atomics, deterministic), the
// per-block partial of dS goes out as one fp32 atomic per (b, j) like the other dS reductions.
__global__ __launch_bounds__(256) void wside_reduce_kernel(const bf16_t* __restrict__ P, const bf16_t* __restrict__ S,
                                                           const bf16_t* __restrict__ Bup, int B, int N, int r,
                                                           float* __restrict__ dB, long lddb, float* __restrict__ dS) {
  constexpr int NR = 16;
  const int n0 = blockIdx.x * NR;
  for (int j = threadIdx.x; j < r; j += blockDim.x) {
    float bup[NR], acc[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int n = n0 + i;
      bup[i] = n < N ? bf16_to_f32(Bup[(long)n * r + j]) : 0.f;
      acc[i] = 0.f;
    }
    for (int b = 0; b < B; ++b) {
      const float s = bf16_to_f32(S[(long)b * r + j]);
      const bf16_t* p = P + ((long)b * N + n0) * r + j;
      float ds = 0.f;
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        const float v = (n0 + i < N) ? bf16_to_f32(p[(long)i * r]) : 0.f;
        acc[i] = fmaf(v, s, acc[i]);
        ds = fmaf(bup[i], v, ds);
      }
      atomicAdd(dS + (long)b * r + j, ds);
    }
#pragma unroll
    for (int i = 0; i < NR; ++i)
      if (n0 + i < N) dB[(long)(n0 + i) * lddb + j] += acc[i];
  }
}

extern "C" int aql_wside_reduce(const bf16_t* P, const bf16_t* S, const bf16_t* Bup, int B, int N, int r, float* dBup, long lddb,
                                float* dS, hipStream_t stream) {
  AQL_CHECK_ARG(P && S && Bup && dBup && dS && B > 0 && N > 0 && r > 0 && lddb >= r, "aql_wside_reduce: bad args");
  hipLaunchKernelGGL(wside_reduce_kernel, dim3((N + 15) / 16), dim3(256), 0, stream, P, S, Bup, B, N, r, dBup, lddb, dS);
  AQL_CHECK_LAUNCH("aql_wside_reduce");
  return AQL_OK;
}

extern "C" int aql_lora_ds(const bf16_t* dTs, const bf16_t* T, int nb, int rows_per_sample, int r, float* dS,
                           hipStream_t stream) {
  AQL_CHECK_ARG(dTs && T && dS && r % 8 == 0 && r <= 1024, "aql_lora_ds: bad args (r=%d)", r);
  const int cols = r / 8;
  const int threads = cols * (256 / cols > 0 ? 256 / cols : 1);
  int slabs = (rows_per_sample + 1023) / 1024;   // every workgroup ends in r atomics onto the SAME [nb][r] accumulator: few, fat slabs
  if (slabs > 32) slabs = 32;
  hipLaunchKernelGGL(lora_ds_kernel, dim3(slabs, nb), dim3(threads), 0, stream, dTs, T, rows_per_sample, r, dS);
  AQL_CHECK_LAUNCH("aql_lora_ds");
  return AQL_OK;
}
extern "C" int aql_ds_desc_fill_ld(void* host_desc, const bf16_t* dTs, long ld_dts, const bf16_t* T, int nb, int rows_per_sample,
                                   int r, float* dS, int first_block);
extern "C" int aql_ds_desc_fill(void* host_desc, const bf16_t* dTs, const bf16_t* T, int nb, int rows_per_sample,
                                int r, float* dS, int first_block) {
  return aql_ds_desc_fill_ld(host_desc, dTs, 0, T, nb, rows_per_sample, r, dS, first_block);
}
// ... with dTs at a row stride of ld_dts elements (0 = dense): a column block of the stacked backward down product of a grouped
// q | k | v (aql_gemm_bf16_grouped)
extern "C" int aql_ds_desc_fill_ld(void* host_desc, const bf16_t* dTs, long ld_dts, const bf16_t* T, int nb, int rows_per_sample,
                                   int r, float* dS, int first_block) {
  if (!host_desc || !dTs || !T || !dS || r % 8 || r > 1024 || r / 8 > 256) return 0;
  if (ld_dts < 0 || ld_dts % 8 || ld_dts > 0x7fffffffL || (ld_dts > 0 && ld_dts < r)) return 0;
  DsGroupDesc d;
  d.dTs = dTs;
  d.T = T;
  d.dS = dS;
  d.nb = nb;
  d.rps = rows_per_sample;
  d.r = r;
  int slabs = (rows_per_sample + 1023) / 1024;   // every workgroup ends in r atomics onto the SAME [nb][r] accumulator: few, fat slabs
  if (slabs > 32) slabs = 32;
  d.slabs = slabs;
  d.first_block = first_block;
  d.pad = (ld_dts == r) ? 0 : (int)ld_dts;
  memcpy(host_desc, &d, sizeof(d));
  return slabs * nb;
}
extern "C" int aql_lora_ds_grouped(const void* dev_descs, int n, int total_blocks, hipStream_t stream) {
  AQL_CHECK_ARG(dev_descs && n > 0 && total_blocks > 0, "aql_lora_ds_grouped: bad args");
  static_assert(sizeof(DsGroupDesc) == 48, "descriptor layout is part of the ABI");
  hipLaunchKernelGGL(lora_ds_grouped_kernel, dim3(total_blocks), dim3(256), 0, stream,
                     static_cast<const DsGroupDesc*>(dev_descs), n);
  AQL_CHECK_LAUNCH("aql_lora_ds_grouped");
  return AQL_OK;
}
extern "C" int aql_sumsq_f32(const float* g, long n, float* out, hipStream_t stream) {
  AQL_CHECK_ARG(g && out, "aql_sumsq_f32: bad args");
  (void)hipMemsetAsync(out, 0, sizeof(float), stream);
  hipLaunchKernelGGL(sumsq_kernel, dim3(grid_for(n, 1024, 1024)), dim3(256), 0, stream, g, n, out);
  AQL_CHECK_LAUNCH("aql_sumsq_f32");
  return AQL_OK;
}
extern "C" int aql_clipnorm_adamw(float* p, const float* g, float* m, float* v, long n, const float* sumsq,
                                  float max_norm, const float* lr, float beta1, float beta2, float eps, float wd,
                                  const int* step, hipStream_t stream) {
  AQL_CHECK_ARG(p && g && m && v && lr && step, "aql_clipnorm_adamw: bad args");
  hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n, 1024, 2048)), dim3(256), 0, stream, p, g, m, v, n, sumsq, max_norm,
                     lr, beta1, beta2, eps, wd, step);
  AQL_CHECK_LAUNCH("aql_clipnorm_adamw");
  return AQL_OK;
}

// ---- DDIM step with classifier-free guidance (diffusers DDIMScheduler.step, eta = 0, epsilon prediction):
//   eps = eps_u + g (eps_c - eps_u);  x0 = (x - sqrt(1-a_t) eps) / sqrt(a_t);  x <- sqrt(a_prev) x0 + sqrt(1-a_prev) eps
// coef (device) = {sqrt(a_t), sqrt(1-a_t), sqrt(a_prev), sqrt(1-a_prev)}; eps_u/eps_c are the two halves of the U-Net output.
namespace {
__global__ __launch_bounds__(256) void ddim_step_kernel(float* __restrict__ x, const bf16_t* __restrict__ eps_u,
                                                        const bf16_t* __restrict__ eps_c, float g,
                                                        const float* __restrict__ coef, long n) {
  const float sa = coef[0], sb = coef[1], pa = coef[2], pb = coef[3];
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < n; id += (long)gridDim.x * blockDim.x) {
    const float eu = bf16_to_f32(eps_u[id]), ec = bf16_to_f32(eps_c[id]);
    const float e = eu + g * (ec - eu);
    const float x0 = (x[id] - sb * e) / sa;
    x[id] = pa * x0 + pb * e;
  }
}
}  // namespace
extern "C" int aql_ddim_step(float* x, const bf16_t* eps_u, const bf16_t* eps_c, float guidance, const float* coef, long n,
                             hipStream_t stream) {
  AQL_CHECK_ARG(x && eps_u && eps_c && coef, "aql_ddim_step: bad args");
  hipLaunchKernelGGL(ddim_step_kernel, dim3(grid_for(n)), dim3(256), 0, stream, x, eps_u, eps_c, guidance, coef, n);
  AQL_CHECK_LAUNCH("aql_ddim_step");
  return AQL_OK;
}

// ---- one phase of a captured sampling loop (evaluation/utils_eval.py:83-102: the schedulers behind `sampler=`) -----------------------
// Every deterministic sampler of the table -- Euler, Heun, KDPM2, LMS, PLMS, DPM-Solver++ single-step, UniPC (and the ancestral
// KDPM2 with caller-supplied noise) -- advances by LINEAR combinations of a handful of buffers between two U-Net calls: the state x,
// one auxiliary state (Heun's / KDPM2's trial point, PLMS's / DPM-Solver's saved sample, UniPC's last sample), up to four history
// entries and the guided noise prediction e = eps_u + g (eps_c - eps_u) that just arrived.  This kernel is that combination with
// everything that varies per phase in DEVICE memory, so that ONE HIP graph [U-Net on the CFG batch, this kernel (twice)] replays for
// every phase of every sampler (aqualora_amd/ksamplers.py writes the per-phase numbers):
//   coef[0] g   coef[1] c_x   coef[2] c_aux   coef[3] c_e   coef[4..7] c_h0..c_h3   coef[8] c_noise   coef[9] next_in_scale
//   coef[10] p_src   coef[11] p_e      (the value pushed into the history: p_src * SRC + p_e * e, SRC = the state the model just saw)
//   flag[0] dst (0 = x, 1 = aux)   flag[1] push (shift h3 <- h2 <- h1 <- h0, h0 <- pushed value, BEFORE the combination)
//   flag[2] next_src (0 = x, 1 = aux: which state the next model call reads)   flag[3] bit 0: no model output in this call (e = 0),
//   bit 1: aux <- x before the combination (save the sample)   flag[4] src (which state the model just saw: 0 = x, 1 = aux)
//   dst <- c_x x + c_aux aux + c_e e + sum_k c_hk h_k + c_noise noise ;   uin[0..n) = uin[n..2n) = next_in_scale * (next source)
namespace {
__global__ __launch_bounds__(256) void sampler_step_kernel(float* __restrict__ x, float* __restrict__ aux, float* __restrict__ hist,
                                                           const float* __restrict__ noise, const bf16_t* __restrict__ eps_u,
                                                           const bf16_t* __restrict__ eps_c, float* __restrict__ uin,
                                                           const float* __restrict__ coef, const int* __restrict__ flag, long n) {
  const float g = coef[0], cx = coef[1], ca = coef[2], ce = coef[3], cn = coef[8], nscale = coef[9], psrc = coef[10], pe = coef[11];
  const float ch0 = coef[4], ch1 = coef[5], ch2 = coef[6], ch3 = coef[7];
  const int dst = flag[0], push = flag[1], nsrc = flag[2], mode = flag[3], src = flag[4];
  const bool no_eval = mode & 1, save = mode & 2;
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < n; id += (long)gridDim.x * blockDim.x) {
    float xv = x[id], av = aux[id];
    float e = 0.f;
    if (!no_eval) {
      const float eu = bf16_to_f32(eps_u[id]), ec = bf16_to_f32(eps_c[id]);
      e = eu + g * (ec - eu);
    }
    float h0 = hist[id], h1 = hist[n + id], h2 = hist[2 * n + id], h3 = hist[3 * n + id];
    if (push) {
      h3 = h2, h2 = h1, h1 = h0;
      h0 = psrc * (src ? av : xv) + pe * e;
      hist[id] = h0, hist[n + id] = h1, hist[2 * n + id] = h2, hist[3 * n + id] = h3;
    }
    if (save) av = xv;
    float v = cx * xv + ca * av + ce * e + ch0 * h0 + ch1 * h1 + ch2 * h2 + ch3 * h3;
    if (cn != 0.f) v += cn * noise[id];
    if (dst) av = v; else xv = v;
    x[id] = xv;
    aux[id] = av;
    const float u = nscale * (nsrc ? av : xv);
    uin[id] = u;
    uin[n + id] = u;
  }
}
}  // namespace
extern "C" int aql_sampler_step(float* x, float* aux, float* hist, const float* noise, const bf16_t* eps_u, const bf16_t* eps_c,
                                float* uin, const float* coef, const int* flag, long n, hipStream_t stream) {
  AQL_CHECK_ARG(x && aux && hist && noise && eps_u && eps_c && uin && coef && flag && n > 0, "aql_sampler_step: bad args");
  hipLaunchKernelGGL(sampler_step_kernel, dim3(grid_for(n)), dim3(256), 0, stream, x, aux, hist, noise, eps_u, eps_c, uin, coef, flag, n);
  AQL_CHECK_LAUNCH("aql_sampler_step");
  return AQL_OK;
}

// ---- row softmax for the VAE's single-head, 512-wide attention (diffusers AutoencoderKL mid-block Attention; the flash
// kernels of aql_attn.hip stop at d = 160): P[m, :] = softmax(scale * S[m, :]) with S fp32 (from aql_gemm_nt_f32_accum)
// and P bf16.  One workgroup per row; the row is read three times (max, sum, write) and stays in L2.
namespace {
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ S, long lds_, int N, float scale,
                                                           bf16_t* __restrict__ P, long ldp) {
  __shared__ float red[4];
  const float* s = S + (long)blockIdx.x * lds_;
  bf16_t* p = P + (long)blockIdx.x * ldp;
  const int tid = threadIdx.x;
  float mx = -INFINITY;
  for (int i = tid; i < N; i += 256) mx = fmaxf(mx, s[i]);
  mx = wave_max(mx);
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
  for (int i = tid; i < N; i += 256) sum += __expf((s[i] - mx) * scale);
  sum = wave_sum(sum);
  if ((tid & 63) == 0) red[tid >> 6] = sum;
  __syncthreads();
  const float inv = 1.f / ((red[0] + red[1]) + (red[2] + red[3]));
  for (int i = tid; i < N; i += 256) p[i] = f32_to_bf16(__expf((s[i] - mx) * scale) * inv);
}
}  // namespace
extern "C" int aql_softmax_rows(const float* S, long lds_, long M, int N, float scale, bf16_t* P, long ldp,
                                hipStream_t stream) {
  AQL_CHECK_ARG(S && P && M > 0 && N > 0 && scale > 0.f, "aql_softmax_rows: bad args");
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)M), dim3(256), 0, stream, S, lds_, N, scale, P, ldp);
  AQL_CHECK_LAUNCH("aql_softmax_rows");
  return AQL_OK;
}

// backward of the row softmax above: dS[m,:] = scale * P[m,:] * (dP[m,:] - sum_j P[m,j] dP[m,j])   (bf16 in, bf16 out)
namespace {
__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(const bf16_t* __restrict__ P, const bf16_t* __restrict__ dP,
                                                               long ld, int N, float scale, bf16_t* __restrict__ dS) {
  __shared__ float red[4];
  const bf16_t* p = P + (long)blockIdx.x * ld;
  const bf16_t* g = dP + (long)blockIdx.x * ld;
  bf16_t* o = dS + (long)blockIdx.x * ld;
  const int tid = threadIdx.x;
  float dot = 0.f;
  for (int i = tid; i < N; i += 256) dot += bf16_to_f32(p[i]) * bf16_to_f32(g[i]);
  dot = wave_sum(dot);
  if ((tid & 63) == 0) red[tid >> 6] = dot;
  __syncthreads();
  const float delta = (red[0] + red[1]) + (red[2] + red[3]);
  for (int i = tid; i < N; i += 256) o[i] = f32_to_bf16(scale * bf16_to_f32(p[i]) * (bf16_to_f32(g[i]) - delta));
}
}  // namespace
extern "C" int aql_softmax_rows_bwd(const bf16_t* P, const bf16_t* dP, long ld, long M, int N, float scale, bf16_t* dS,
                                    hipStream_t stream) {
  AQL_CHECK_ARG(P && dP && dS && M > 0 && N > 0, "aql_softmax_rows_bwd: bad args");
  hipLaunchKernelGGL(softmax_rows_bwd_kernel, dim3((unsigned)M), dim3(256), 0, stream, P, dP, ld, N, scale, dS);
  AQL_CHECK_LAUNCH("aql_softmax_rows_bwd");
  return AQL_OK;
}

// ---- CLIP text encoder pieces (transformers CLIPTextModel, reference call site train/ppft_train.py:1014-1019) ---------
// quick_gelu(x) = x * sigmoid(1.702 x)   (CLIPMLP activation), bf16 in/out
namespace {
__global__ __launch_bounds__(256) void quick_gelu_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, long n8) {
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < n8; id += (long)gridDim.x * blockDim.x) {
    const uint4 v = reinterpret_cast<const uint4*>(x)[id];
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float a = bf16lo(w[i]), b = bf16hi(w[i]);
      o[i] = pack_bf16x2(a / (1.f + __expf(-1.702f * a)), b / (1.f + __expf(-1.702f * b)));
    }
    reinterpret_cast<uint4*>(y)[id] = make_uint4(o[0], o[1], o[2], o[3]);
  }
}

// Causal self-attention for short sequences (N <= 128, d <= 128; CLIP text: N = 77, 12 heads of 64): one workgroup per
// (head, sample), K and V of the head in LDS as fp32, one wavefront per query row: lanes over keys for the scores and
// the softmax, lanes over channels for P.V.  q/k/v/o: [B, N, H*d] bf16 (head h owns columns h*d..), row stride ld.
constexpr int CA_MAXN = 128, CA_MAXD = 128;
__global__ __launch_bounds__(256) void causal_attn_small_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                                const bf16_t* __restrict__ v, long ld, int N, int d,
                                                                float scale, bf16_t* __restrict__ o, long ldo) {
  extern __shared__ float sm[];  // K [N][d+1], V [N][d+1], P [4][CA_MAXN], Q [4][CA_MAXD]
  const int pitch = d + 1;
  float* sK = sm;
  float* sV = sK + N * pitch;
  float* sP = sV + N * pitch;
  float* sQ = sP + 4 * CA_MAXN;
  const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long base = (long)b * N * ld + (long)h * d;
  for (int i = tid; i < N * d; i += 256) {
    const int r = i / d, c = i - r * d;
    sK[r * pitch + c] = bf16_to_f32(k[base + (long)r * ld + c]);
    sV[r * pitch + c] = bf16_to_f32(v[base + (long)r * ld + c]);
  }
  __syncthreads();
  for (int i = wave; i < N; i += 4) {
    for (int c = lane; c < d; c += 64) sQ[wave * CA_MAXD + c] = bf16_to_f32(q[base + (long)i * ld + c]) * scale;
    __builtin_amdgcn_wave_barrier();
    float s[2], mx = -INFINITY;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int j = lane + 64 * u;
      float acc = -INFINITY;
      if (j <= i) {  // causal: key j may be seen by query i iff j <= i
        acc = 0.f;
        for (int c = 0; c < d; ++c) acc += sQ[wave * CA_MAXD + c] * sK[j * pitch + c];
      }
      s[u] = acc;
      mx = fmaxf(mx, acc);
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int j = lane + 64 * u;
      const float p = (j <= i) ? __expf(s[u] - mx) : 0.f;
      if (j < CA_MAXN) sP[wave * CA_MAXN + j] = p;
      sum += p;
    }
    sum = wave_sum(sum);
    __builtin_amdgcn_wave_barrier();
    const float inv = 1.f / sum;
    for (int c = lane; c < d; c += 64) {
      float acc = 0.f;
      for (int j = 0; j <= i; ++j) acc += sP[wave * CA_MAXN + j] * sV[j * pitch + c];
      o[(long)b * N * ldo + (long)i * ldo + (long)h * d + c] = f32_to_bf16(acc * inv);
    }
    __builtin_amdgcn_wave_barrier();
  }
}
}  // namespace

extern "C" int aql_quick_gelu(const bf16_t* x, long n, bf16_t* y, hipStream_t stream) {
  AQL_CHECK_ARG(x && y && n > 0 && n % 8 == 0, "aql_quick_gelu: n must be a positive multiple of 8");
  hipLaunchKernelGGL(quick_gelu_kernel, dim3(grid_for(n / 8)), dim3(256), 0, stream, x, y, n / 8);
  AQL_CHECK_LAUNCH("aql_quick_gelu");
  return AQL_OK;
}

extern "C" int aql_causal_attn_small(const bf16_t* q, const bf16_t* k, const bf16_t* v, long ld, int B, int H, int N, int d,
                                     float scale, bf16_t* o, long ldo, hipStream_t stream) {
  AQL_CHECK_ARG(q && k && v && o, "aql_causal_attn_small: null operand");
  AQL_CHECK_ARG(N > 0 && N <= CA_MAXN && d > 0 && d <= CA_MAXD, "aql_causal_attn_small: N <= 128 and d <= 128 (N=%d d=%d)", N, d);
  const size_t lds = (size_t)(2 * N * (d + 1) + 4 * CA_MAXN + 4 * CA_MAXD) * sizeof(float);
  AQL_CHECK_ARG(lds <= 64 * 1024, "aql_causal_attn_small: K/V of one head must fit 64 KB of LDS (N=%d d=%d)", N, d);
  hipLaunchKernelGGL(causal_attn_small_kernel, dim3(H, B), dim3(256), lds, stream, q, k, v, ld, N, d, scale, o, ldo);
  AQL_CHECK_LAUNCH("aql_causal_attn_small");
  return AQL_OK;
}

// ---- DPM-Solver++ (2M, data prediction, epsilon model) step with classifier-free guidance: the sampler that
// train/rob_enhance_finetune.py:993,1012 puts in front of the decoder fine-tune (DPMSolverMultistepScheduler, 20 steps).
//   eps = eps_u + g (eps_c - eps_u);  x0 = (x - sigma_t eps) / alpha_t
//   x <- a x + b x0 + c x0_prev ;  x0_prev <- x0        coef (device) = {alpha_t, sigma_t, a, b, c}
// (first-order steps pass c = 0; the host derives a, b, c from the lambda schedule, see inference.dpm_solver_sample)
namespace {
__global__ __launch_bounds__(256) void dpmpp2m_step_kernel(float* __restrict__ x, const bf16_t* __restrict__ eps_u,
                                                           const bf16_t* __restrict__ eps_c, float g,
                                                           float* __restrict__ x0_prev, const float* __restrict__ coef,
                                                           long n) {
  const float al = coef[0], sg = coef[1], a = coef[2], b = coef[3], c = coef[4];
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < n; id += (long)gridDim.x * blockDim.x) {
    const float eu = bf16_to_f32(eps_u[id]), ec = bf16_to_f32(eps_c[id]);
    const float e = eu + g * (ec - eu);
    const float x0 = (x[id] - sg * e) / al;
    x[id] = a * x[id] + b * x0 + c * x0_prev[id];
    x0_prev[id] = x0;
  }
}
}  // namespace
extern "C" int aql_dpmpp2m_step(float* x, const bf16_t* eps_u, const bf16_t* eps_c, float guidance, float* x0_prev,
                                const float* coef, long n, hipStream_t stream) {
  AQL_CHECK_ARG(x && eps_u && eps_c && x0_prev && coef, "aql_dpmpp2m_step: bad args");
  hipLaunchKernelGGL(dpmpp2m_step_kernel, dim3(grid_for(n)), dim3(256), 0, stream, x, eps_u, eps_c, guidance, x0_prev, coef, n);
  AQL_CHECK_LAUNCH("aql_dpmpp2m_step");
  return AQL_OK;
}
