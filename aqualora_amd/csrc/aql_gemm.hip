// C-ABI entry points built on the MFMA GEMM core (aql_gemm.cuh).  See include/aqualora_hip.h for the
// contract of every symbol and the reference interface (file:line) it replaces.
#include "aql_gemm.cuh"
#include <stdarg.h>

using namespace aqlgemm;

static thread_local char g_err[512] = "";
extern "C" const char* aql_last_error(void) { return g_err; }
void aql_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

namespace {

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Sum split-K slabs and apply the bf16 epilogue (bias, per-sample row bias, residual).
__global__ __launch_bounds__(256) void splitk_finalize_kernel(const float* __restrict__ slabs, int splits, long M,
                                                              int N, const bf16_t* __restrict__ bias,
                                                              const bf16_t* __restrict__ rowbias,
                                                              int rows_per_sample,
                                                              const bf16_t* __restrict__ residual, long ldr,
                                                              bf16_t* __restrict__ C, long ldc) {
  const long nchunk = M * (N / 4);
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < nchunk; id += (long)gridDim.x * blockDim.x) {
    const long m = id / (N / 4);
    const int n = (int)(id - m * (N / 4)) * 4;
    float4 s = *reinterpret_cast<const float4*>(slabs + m * N + n);
    for (int z = 1; z < splits; ++z) {
      const float4 t = *reinterpret_cast<const float4*>(slabs + ((long)z * M + m) * N + n);
      s.x += t.x;
      s.y += t.y;
      s.z += t.z;
      s.w += t.w;
    }
    if (bias != nullptr) {
      const uint2 b = *reinterpret_cast<const uint2*>(bias + n);
      s.x += bf16lo(b.x);
      s.y += bf16hi(b.x);
      s.z += bf16lo(b.y);
      s.w += bf16hi(b.y);
    }
    uint2 v = make_uint2(pack_bf16x2(s.x, s.y), pack_bf16x2(s.z, s.w));
    if (rowbias != nullptr) {
      const uint2 r = *reinterpret_cast<const uint2*>(rowbias + (m / rows_per_sample) * N + n);
      v.x = pack_bf16x2(bf16lo(v.x) + bf16lo(r.x), bf16hi(v.x) + bf16hi(r.x));
      v.y = pack_bf16x2(bf16lo(v.y) + bf16lo(r.y), bf16hi(v.y) + bf16hi(r.y));
    }
    if (residual != nullptr) {
      const uint2 r = *reinterpret_cast<const uint2*>(residual + m * ldr + n);
      v.x = pack_bf16x2(bf16lo(v.x) + bf16lo(r.x), bf16hi(v.x) + bf16hi(r.x));
      v.y = pack_bf16x2(bf16lo(v.y) + bf16lo(r.y), bf16hi(v.y) + bf16hi(r.y));
    }
    *reinterpret_cast<uint2*>(C + m * ldc + n) = v;
  }
}

struct OutSpec {
  const bf16_t* bias;
  const bf16_t* rowbias;
  int rows_per_sample;
  const bf16_t* residual;
  long ldr;
  bf16_t* C;
  long ldc;
  bf16_t* C2;
  long ldc2;
  const bf16_t* rowscale;
};

inline int pick_splits(int tiles, int kt_total, long M, int N, size_t ws_bytes) {
  if (tiles >= 192 || kt_total < 8) return 1;
  int s = (384 + tiles - 1) / tiles;
  if (s > kt_total / 4) s = kt_total / 4;
  if (s > 32) s = 32;
  while (s > 1 && (size_t)s * (size_t)M * (size_t)N * 4u > ws_bytes) --s;
  return s < 1 ? 1 : s;
}

// Dispatch one bf16-output GEMM over the tile configurations; falls back to split-K slabs + finalize when the
// tile count cannot fill 256 CUs and the caller supplied a workspace.
template <class LA, class LB>
int run_bf16_gemm(GemmArgs<LA, LB> g, const OutSpec& o, float* ws, size_t ws_bytes, hipStream_t stream,
                  const char* name) {
  g.epi = EpiParams{};
  g.epi.rows_per_sample = o.rows_per_sample > 0 ? o.rows_per_sample : 1;
  const int kt_total = g.ktiles0 + g.ktiles1;
  const bool narrow = (g.N <= 32);
  const bool n64 = !narrow && (g.N % 128 != 0) && (g.N % 128 <= 64);
  const int BNsel = narrow ? 32 : (n64 ? 64 : 128);
  const int tiles = aql_cdiv(g.M, 128) * aql_cdiv(g.N, BNsel);
  int splits = 1;
  if (ws != nullptr && o.C2 == nullptr) splits = pick_splits(tiles, kt_total, g.M, g.N, ws_bytes);
  const bool slab = (splits > 1);
  g.splits = splits;
  if (slab) {
    g.epi.Cf = ws;
    g.epi.ldcf = g.N;
    if (narrow)
      launch_gemm<128, 32, 32, 32, LA, LB, EPI_SLAB>(g, stream);
    else if (n64)
      launch_gemm<128, 64, 64, 32, LA, LB, EPI_SLAB>(g, stream);
    else
      launch_gemm<128, 128, 64, 64, LA, LB, EPI_SLAB>(g, stream);
    AQL_CHECK_LAUNCH(name);
    const long nchunk = (long)g.M * (g.N / 4);
    int blocks = (int)((nchunk + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(splitk_finalize_kernel, dim3(blocks), dim3(256), 0, stream, ws, splits, (long)g.M, g.N,
                       o.bias, o.rowbias, g.epi.rows_per_sample, o.residual, o.ldr, o.C, o.ldc);
    AQL_CHECK_LAUNCH(name);
    return AQL_OK;
  }
  g.epi.C = o.C;
  g.epi.ldc = o.ldc;
  g.epi.bias = o.bias;
  g.epi.residual = o.residual;
  g.epi.ldr = o.ldr;
  g.epi.C2 = o.C2;
  g.epi.ldc2 = o.ldc2;
  g.epi.rowscale = o.rowscale;
  g.epi.rowbias = o.rowbias;
  if (narrow)
    launch_gemm<128, 32, 32, 32, LA, LB, EPI_BF16>(g, stream);
  else if (n64)
    launch_gemm<128, 64, 64, 32, LA, LB, EPI_BF16>(g, stream);
  else
    launch_gemm<128, 128, 64, 64, LA, LB, EPI_BF16>(g, stream);
  AQL_CHECK_LAUNCH(name);
  return AQL_OK;
}

inline PlainLoader plain(const bf16_t* p, long ld, long rows, int K) {
  PlainLoader l;
  l.base = p;
  l.ld = ld;
  l.rows = (int)rows;
  l.K = K;
  return l;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
extern "C" int aql_gemm_bf16(const bf16_t* A, long lda, const bf16_t* B, long ldb, long M, int N, int K,
                             const bf16_t* A2, long lda2, const bf16_t* B2, long ldb2, int K2, const bf16_t* bias,
                             const bf16_t* rowbias, int rows_per_sample, const bf16_t* residual, long ldr,
                             bf16_t* C, long ldc, float* ws, size_t ws_bytes, hipStream_t stream) {
  AQL_CHECK_ARG(A && B && C, "aql_gemm_bf16: null operand");
  AQL_CHECK_ARG(M > 0 && N > 0 && K > 0 && M < (1L << 31), "aql_gemm_bf16: bad shape M=%ld N=%d K=%d", M, N, K);
  AQL_CHECK_ARG(N % 8 == 0 && K % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0,
                "aql_gemm_bf16: N,K and leading dims must be multiples of 8 (N=%d K=%d)", N, K);
  AQL_CHECK_ARG(aligned16(A) && aligned16(B) && aligned16(C), "aql_gemm_bf16: pointers must be 16-byte aligned");
  AQL_CHECK_ARG(residual == nullptr || (ldr % 8 == 0 && aligned16(residual)), "aql_gemm_bf16: bad residual");
  GemmArgs<PlainLoader, PlainLoader> g;
  g.a0 = plain(A, lda, M, K);
  g.b0 = plain(B, ldb, N, K);
  g.ktiles0 = aql_cdiv(K, BK);
  g.ktiles1 = 0;
  g.a1 = g.a0;
  g.b1 = g.b0;
  if (A2 != nullptr) {
    AQL_CHECK_ARG(B2 && K2 > 0 && K2 % 8 == 0 && lda2 % 8 == 0 && ldb2 % 8 == 0 && aligned16(A2) && aligned16(B2),
                  "aql_gemm_bf16: bad second K segment");
    g.a1 = plain(A2, lda2, M, K2);
    g.b1 = plain(B2, ldb2, N, K2);
    g.ktiles1 = aql_cdiv(K2, BK);
  }
  g.M = (int)M;
  g.N = N;
  OutSpec o{bias, rowbias, rows_per_sample, residual, ldr, C, ldc, nullptr, 0, nullptr};
  return run_bf16_gemm(g, o, ws, ws_bytes, stream, "aql_gemm_bf16");
}

// T = X.Adown^T (bf16) and Ts = T * S[sample]  -- the rank-r "down" half of the watermark LoRA.
extern "C" int aql_lora_down(const bf16_t* X, long ldx, long M, int K, const bf16_t* Adown, int r, const bf16_t* S,
                             int rows_per_sample, bf16_t* T, bf16_t* Ts, hipStream_t stream) {
  AQL_CHECK_ARG(X && Adown && S && T && Ts, "aql_lora_down: null operand");
  AQL_CHECK_ARG(M > 0 && r > 0 && r % 8 == 0 && K % 8 == 0 && ldx % 8 == 0 && rows_per_sample > 0,
                "aql_lora_down: bad shape M=%ld r=%d K=%d", M, r, K);
  GemmArgs<PlainLoader, PlainLoader> g;
  g.a0 = plain(X, ldx, M, K);
  g.b0 = plain(Adown, K, r, K);
  g.a1 = g.a0;
  g.b1 = g.b0;
  g.ktiles0 = aql_cdiv(K, BK);
  g.ktiles1 = 0;
  g.M = (int)M;
  g.N = r;
  OutSpec o{nullptr, nullptr, rows_per_sample, nullptr, 0, T, r, Ts, r, S};
  return run_bf16_gemm(g, o, nullptr, 0, stream, "aql_lora_down");
}

extern "C" int aql_conv3x3_fwd(const bf16_t* X, int B, int Hin, int Win, int Cin, const bf16_t* Wk, const bf16_t* bias,
                               int Cout, int stride, int upsample, const bf16_t* rowbias, const bf16_t* residual,
                               bf16_t* Y, float* ws, size_t ws_bytes, hipStream_t stream) {
  AQL_CHECK_ARG(X && Wk && Y, "aql_conv3x3_fwd: null operand");
  AQL_CHECK_ARG(Cin % 8 == 0 && Cout % 8 == 0, "aql_conv3x3_fwd: Cin/Cout must be multiples of 8 (%d,%d)", Cin, Cout);
  AQL_CHECK_ARG((stride == 1 || stride == 2) && (upsample == 0 || upsample == 1) && !(upsample && stride == 2),
                "aql_conv3x3_fwd: bad stride/upsample");
  ConvFwdLoader l;
  l.base = X;
  l.B = B;
  l.Hin = Hin;
  l.Win = Win;
  l.Cin = Cin;
  l.stride = stride;
  l.ups = upsample;
  const int Hl = Hin << upsample, Wl = Win << upsample;
  l.Hout = (Hl + 2 - 3) / stride + 1;
  l.Wout = (Wl + 2 - 3) / stride + 1;
  l.rows = B * l.Hout * l.Wout;
  l.K = 9 * Cin;
  GemmArgs<ConvFwdLoader, PlainLoader> g;
  g.a0 = l;
  g.a1 = l;
  g.b0 = plain(Wk, 9L * Cin, Cout, 9 * Cin);
  g.b1 = g.b0;
  g.ktiles0 = aql_cdiv(9 * Cin, BK);
  g.ktiles1 = 0;
  g.M = l.rows;
  g.N = Cout;
  OutSpec o{bias, rowbias, l.Hout * l.Wout, residual, Cout, Y, Cout, nullptr, 0, nullptr};
  return run_bf16_gemm(g, o, ws, ws_bytes, stream, "aql_conv3x3_fwd");
}

// dX[b,hi,wi,ci] = sum_{kh,kw,co} dY[b,ho,wo,co] * Wt[ci][(kh*3+kw)*Cout+co],  hi = ho*stride + kh - 1.
extern "C" int aql_conv3x3_bwd_data(const bf16_t* dY, int B, int Hin, int Win, int Cin, const bf16_t* Wt, int Cout,
                                    int stride, bf16_t* dX, float* ws, size_t ws_bytes, hipStream_t stream) {
  AQL_CHECK_ARG(dY && Wt && dX, "aql_conv3x3_bwd_data: null operand");
  AQL_CHECK_ARG(Cin % 8 == 0 && Cout % 8 == 0 && (stride == 1 || stride == 2), "aql_conv3x3_bwd_data: bad shape");
  ConvBwdLoader l;
  l.base = dY;
  l.B = B;
  l.Hin = Hin;
  l.Win = Win;
  l.Cout = Cout;
  l.stride = stride;
  l.Hout = (Hin + 2 - 3) / stride + 1;
  l.Wout = (Win + 2 - 3) / stride + 1;
  l.rows = B * Hin * Win;
  l.K = 9 * Cout;
  GemmArgs<ConvBwdLoader, PlainLoader> g;
  g.a0 = l;
  g.a1 = l;
  g.b0 = plain(Wt, 9L * Cout, Cin, 9 * Cout);
  g.b1 = g.b0;
  g.ktiles0 = aql_cdiv(9 * Cout, BK);
  g.ktiles1 = 0;
  g.M = l.rows;
  g.N = Cin;
  OutSpec o{nullptr, nullptr, 1, nullptr, 0, dX, Cin, nullptr, 0, nullptr};
  return run_bf16_gemm(g, o, ws, ws_bytes, stream, "aql_conv3x3_bwd_data");
}

// C[P,Q] (fp32) += alpha * sum_m U[m,P] * V[m,Q]   -- token-reduction GEMM for the LoRA weight gradients.
// The reduction over m is split across workgroups and combined with fp32 atomics, so C must hold the value to
// accumulate onto (zero for a fresh gradient).
extern "C" int aql_gemm_tn_f32(const bf16_t* U, long ldu, const bf16_t* V, long ldv, long M, int P, int Q, float alpha,
                               float* C, long ldc, hipStream_t stream) {
  AQL_CHECK_ARG(U && V && C, "aql_gemm_tn_f32: null operand");
  AQL_CHECK_ARG(M > 0 && P % 8 == 0 && Q % 8 == 0 && ldu % 8 == 0 && ldv % 8 == 0 && M < (1L << 31),
                "aql_gemm_tn_f32: bad shape M=%ld P=%d Q=%d", M, P, Q);
  GemmArgs<TransLoader, TransLoader> g;
  g.a0.base = U;
  g.a0.ld = ldu;
  g.a0.rows = P;
  g.a0.K = (int)M;
  g.b0.base = V;
  g.b0.ld = ldv;
  g.b0.rows = Q;
  g.b0.K = (int)M;
  g.a1 = g.a0;
  g.b1 = g.b0;
  g.ktiles0 = aql_cdiv(M, BK);
  g.ktiles1 = 0;
  g.M = P;
  g.N = Q;
  g.epi = EpiParams{};
  g.epi.Cf = C;
  g.epi.ldcf = ldc;
  g.epi.alpha = alpha;
  const bool narrow = (Q <= 32);
  const bool n64 = !narrow && (Q % 128 != 0) && (Q % 128 <= 64);
  const int tiles = aql_cdiv(P, 128) * aql_cdiv(Q, narrow ? 32 : (n64 ? 64 : 128));
  int splits = (512 + tiles - 1) / tiles;
  if (splits > g.ktiles0) splits = g.ktiles0;
  if (splits < 1) splits = 1;
  g.splits = splits;
  if (narrow)
    launch_gemm<128, 32, 32, 32, TransLoader, TransLoader, EPI_ATOMIC>(g, stream);
  else if (n64)
    launch_gemm<128, 64, 64, 32, TransLoader, TransLoader, EPI_ATOMIC>(g, stream);
  else
    launch_gemm<128, 128, 64, 64, TransLoader, TransLoader, EPI_ATOMIC>(g, stream);
  AQL_CHECK_LAUNCH("aql_gemm_tn_f32");
  return AQL_OK;
}
