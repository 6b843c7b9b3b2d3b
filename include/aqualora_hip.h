/* aqualora_hip -- C ABI of the MI355X (gfx950) kernels behind AquaLoRA's PPFT / latent-watermark hot path.
 *
 * Drop-in boundary.  The reference has no FFI: its "operator API" is a set of Python forwards
 * monkey-patched onto diffusers modules.  Each entry point below names the reference interface it replaces
 * (paths relative to the reference tree).  Conventions:
 *   - every pointer is a DEVICE pointer owned by the caller (including workspaces); nothing is allocated inside;
 *   - bf16_t is raw bfloat16 bits; activations are row-major with channels last ([tokens, C] / [B,H,W,C]);
 *   - pointers are 16-byte aligned, channel counts / leading dims are multiples of 8;
 *   - `stream` is a hipStream_t; calls are asynchronous, stateless and re-entrant (safe from autograd threads);
 *   - return 0 on success, non-zero otherwise with a message in aql_last_error() (thread-local).
 * Python binding: aqualora_amd/_lib.py (ctypes); see INTEGRATION.md for the reference-side stubs.
 */
#ifndef AQUALORA_HIP_H
#define AQUALORA_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef uint16_t bf16_t;
typedef void* aql_stream_t; /* hipStream_t */

const char* aql_last_error(void);

/* ---- GEMM family (csrc/aql_gemm.hip) ---------------------------------------------------------------------------
 * C[M,N] = A[M,K].B[N,K]^T (+ A2[M,K2].B2[N,K2]^T) + bias[N] + rowbias[m / rows_per_sample][N] + residual[M,N]
 * One MFMA accumulator for both K segments: this IS the fused watermark-LoRA linear
 *   nn.Linear.forward(x) + lora_layer(x, scale)          utils/lora_modules.py:56-62 (and :46-54 for 1x1 convs)
 * with A2 = (x.A^T)*S from aql_lora_down and B2 = up.weight.  ws: optional fp32 split-K workspace.             */
int aql_gemm_bf16(const bf16_t* A, long lda, const bf16_t* B, long ldb, long M, int N, int K, const bf16_t* A2,
                  long lda2, const bf16_t* B2, long ldb2, int K2, const bf16_t* bias, const bf16_t* rowbias,
                  int rows_per_sample, const bf16_t* residual, long ldr, bf16_t* C, long ldc, float* ws,
                  size_t ws_bytes, aql_stream_t stream);
/* Twin batches (clean samples of ppft_train.py:1026-1029 in rows [0, lora_row0), watermarked samples after them, one launch
 * for both passes): rows below lora_row0 carry an all-zero scale, i.e. no LoRA term -- tiles that end at or below it skip the
 * second K segment, straddling tiles read those A2 rows as zeros.  The same trailing `lora_row0` argument of the one-launch
 * entry points below additionally skips the A-tile traffic, the T product and the T / Ts (and GEGLU pre-activation) output of
 * such tiles.  lora_row0 = 0: every row has a LoRA term.                                                                */
int aql_gemm_bf16_ex(const bf16_t* A, long lda, const bf16_t* B, long ldb, long M, int N, int K, const bf16_t* A2,
                     long lda2, const bf16_t* B2, long ldb2, int K2, const bf16_t* bias, const bf16_t* rowbias,
                     int rows_per_sample, const bf16_t* residual, long ldr, bf16_t* C, long ldc, long lora_row0, float* ws,
                     size_t ws_bytes, aql_stream_t stream);

/* The same GEMM with PER-SAMPLE weights: the weight-side form of the watermark-LoRA linear (utils/lora_modules.py:9-26,56-62) for
 * ranks that are not small against the channel count (rank 320 on 320 channels, BASELINE config 3):
 *   C[m][:] = A[m][:] . Wsel(m)^T (+ bias) (+ residual[res_mod ? m % res_mod : m]),
 *   Wsel(m) = B for m < srow0, else Bs + ((m - srow0) / srows) * sstride       (We_b = W + Bup.diag(S_b).A, built by this entry too:
 *   A = stacked scaled up-matrices [(b,n)][r], B = A_down^T, residual = W with res_mod = N).  srows % 256 == 0.
 * geglu_F > 0: the ff.net.0 form of aql_gemm_bf16_geglu (C = H may be null, written for rows >= c_row0); gb_h: the GEGLU-backward
 * epilogue of aql_gemm_bf16_geglu_bwd.  Returns 100 where the GEGLU forms have no tile.                                          */
int aql_gemm_bf16_sw(const bf16_t* A, long lda, const bf16_t* B, long ldb, long M, int N, int K, const bf16_t* Bs, long sstride,
                     long srows, long srow0, const bf16_t* bias, const bf16_t* residual, long ldr, long res_mod, bf16_t* C,
                     long ldc, bf16_t* G, long ldg, int geglu_F, long c_row0, const bf16_t* gb_h, long gb_ldh, float* ws,
                     size_t ws_bytes, aql_stream_t stream);

/* T = X.Adown^T ; Ts = T * S[m / rows_per_sample]   ==  down(x) @ diag_embed(scale)
 *   utils/lora_modules.py:13-17 (linear) and :33-36 (conv, scale[:, :, None, None]).                           */
/* Backward reuse: with X = dY, Adown = up.weight^T it yields dTs and dT = dTs*S; passing Tref (the forward T) and dS
 * additionally accumulates dS[sample,:] += sum_rows dTs*Tref, the gradient of the diagonal.                       */
int aql_lora_down(const bf16_t* X, long ldx, long M, int K, const bf16_t* Adown, int r, const bf16_t* S,
                  int rows_per_sample, bf16_t* T, bf16_t* Ts, const bf16_t* Tref, float* dS, aql_stream_t stream);
/* The same product (utils/lora_modules.py:13-17) for few rows under a deep contraction (M <= ~2048, K >= 2048: the
 * backward-data pass of ff.net.0 at the 16x16 / 8x8 levels): the K range is cut over workgroups, fp32 partials in `part`
 * (>= 32 * M * 32 floats covers every split), the last arrival of a 16-row block (ticket in `counters`: >= ceil(M/16) ints,
 * zero before the call, left zero) adds them in piece order -- deterministic.  Calls aql_lora_down when no split pays.  */
int aql_lora_down_splitk(const bf16_t* X, long ldx, long M, int K, const bf16_t* Adown, int r, const bf16_t* S,
                         int rows_per_sample, bf16_t* T, bf16_t* Ts, float* part, size_t part_bytes, int* counters,
                         size_t counters_bytes, aql_stream_t stream);

/* 3x3 convolution, pad 1, stride 1|2, NHWC, weights Wk[Cout][(kh*3+kw)*Cin+ci]; upsample=1 folds the nearest x2
 * of Upsample2D into the gather.  Replaces F.conv2d in CustomLoRACompatibleConvforward (lora_modules.py:47-52)
 * for the ResNet / down / up-sampler convs (scripts/lib/original_unet.py:425-430, 534, 1058).                    */
int aql_conv3x3_fwd(const bf16_t* X, int B, int Hin, int Win, int Cin, const bf16_t* Wk, const bf16_t* bias, int Cout,
                    int stride, int upsample, const bf16_t* rowbias, long rowbias_ld, const bf16_t* residual, bf16_t* Y,
                    float* ws, size_t ws_bytes, aql_stream_t stream);
/* its input gradient (autograd of the same call); Wt[Cin][(kh*3+kw)*Cout+co]                                     */
/* pad_lo = 1: identical to aql_conv3x3_fwd.  pad_lo = 0 (stride 2, even H/W): F.pad(x,(0,1,0,1)) + Conv2d(3, stride 2,
 * padding 0), the Downsample2D of diffusers' AutoencoderKL encoder (frozen VAE encode, train/ppft_train.py:993).  */
int aql_conv3x3_fwd_pad(const bf16_t* X, int B, int Hin, int Win, int Cin, const bf16_t* Wk, const bf16_t* bias, int Cout,
                        int stride, int upsample, int pad_lo, const bf16_t* rowbias, long rowbias_ld,
                        const bf16_t* residual, bf16_t* Y, float* ws, size_t ws_bytes, aql_stream_t stream);
/* ONE launch for the whole rank-32 LoRA linear (lora_modules.py:9-26 + 56-62):
 *   T = X.A^T, Ts = T * S[m / rows_per_sample], Y = X.W^T + Ts.Bup^T + bias + residual      (A [32,K], Bup [N,32])
 * T, Ts [M,32] are written for the backward pass.  With (X, W, A, Bup) := (dY, W^T, Bup^T, A^T) it is the backward-data
 * form dTs = dY.Bup, dT = dTs * S, dX = dY.W + dT.A.  Returns 100 (not an error) when the shape belongs on the two-launch
 * path aql_lora_down + aql_gemm_bf16 (split-K shapes, N <= 32).                                                      */
int aql_lora_gemm_fused(const bf16_t* X, long ldx, const bf16_t* W, long ldw, long M, int N, int K, const bf16_t* Adown,
                        const bf16_t* S, int rows_per_sample, const bf16_t* Bup, const bf16_t* bias,
                        const bf16_t* residual, long ldr, bf16_t* Y, long ldy, bf16_t* T, bf16_t* Ts, long lora_row0,
                        aql_stream_t stream);
/* ff.net.0.proj + GEGLU in ONE launch (GEGLU.forward, scripts/lib/original_unet.py:727-729 -- diffusers' FeedForward --
 * on top of lora_modules.py:56-62 / 9-26):  h = X.W^T (+ LoRA) + bias with W [2F,K];  G = h[:, :F] * gelu_erf(h[:, F:]).
 * Each 160-wide output tile holds 80 value columns and their 80 gate columns, so the activation runs in the epilogue on the
 * bf16-rounded tile (bit-identical to the GEMM followed by aql_geglu_fwd).  H [M,2F] (the pre-activation that
 * aql_geglu_bwd needs) is written only when non-null: the frozen pass skips it.  Both return 100 (not an error) when the
 * shape has no 160-wide tile (F % 80 != 0, tiny grids); aql_gemm_bf16_geglu takes the optional second K segment of
 * aql_gemm_bf16 (the two-launch LoRA form for ranks other than 32).                                                    */
int aql_gemm_bf16_geglu(const bf16_t* A, long lda, const bf16_t* B, long ldb, long M, int F, int K, const bf16_t* A2,
                        long lda2, const bf16_t* B2, long ldb2, int K2, const bf16_t* bias, bf16_t* H, long ldh, bf16_t* G,
                        long ldg, long lora_row0, aql_stream_t stream);
int aql_lora_gemm_fused_geglu(const bf16_t* X, long ldx, const bf16_t* W, long ldw, long M, int F, int K,
                              const bf16_t* Adown, const bf16_t* S, int rows_per_sample, const bf16_t* Bup,
                              const bf16_t* bias, bf16_t* H, long ldh, bf16_t* G, long ldg, bf16_t* T, bf16_t* Ts,
                              long lora_row0, aql_stream_t stream);
/* Several rank-32 LoRA linears that share their input as ONE launch of aql_lora_gemm_fused: q|k|v of a self-attention
 * (attn.to_q / to_k / to_v, scripts/lib/original_unet.py:688-704) or the k|v projections of the text states of all 16
 * cross-attentions.  W [N,K], Bup [N,32], bias [N] are the linears stacked along N; Adown [ngroups*32, K] the stacked down
 * matrices; T, Ts [ngroups][M][32].  col_start[0..ngroups] (HOST array): first output column of each linear (multiples of
 * 160; col_start[ngroups] = N).  Returns 100 like aql_lora_gemm_fused.                                                  */
int aql_lora_gemm_fused_grouped(const bf16_t* X, long ldx, const bf16_t* W, long ldw, long M, int N, int K, int ngroups,
                                const int* col_start, const bf16_t* Adown, const bf16_t* S, int rows_per_sample,
                                const bf16_t* Bup, const bf16_t* bias, bf16_t* Y, long ldy, bf16_t* T, bf16_t* Ts,
                                long lora_row0, aql_stream_t stream);
/* Backward-data of ff.net.2 (aql_lora_gemm_fused with X = dY, W = W2^T, Adown = Bup2^T, Bup = A2^T) fused with the backward
 * of the GEGLU in front of it (original_unet.py:727-729): the epilogue turns d(value * gelu(gate)) [M,F] into d(pre-activation)
 * DH [M,2F] using the saved pre-activation H [M,2F]; T, Ts receive dTs, dT.  Returns 100 like aql_lora_gemm_fused.        */
int aql_lora_gemm_fused_geglu_bwd(const bf16_t* X, long ldx, const bf16_t* W, long ldw, long M, int F, int K,
                                  const bf16_t* Adown, const bf16_t* S, int rows_per_sample, const bf16_t* Bup,
                                  const bf16_t* H, long ldh, bf16_t* DH, long lddh, bf16_t* T, bf16_t* Ts,
                                  aql_stream_t stream);
/* The same fusion for the two-launch LoRA form (any rank; configs 3 / 5 train at rank 320, train/README.md:34-48):
 * d(activated) [M,F] = A.B^T + A2.B2^T (A = dY, B = W2^T, A2 = dT, B2 = A2^T of the ff.net.2 site), written as d(pre-activation)
 * DH [M,2F] through the GEGLU backward (original_unet.py:727-729) of the saved pre-activation H [M,2F].  Returns 100 when the shape
 * takes the split-K path (caller: aql_gemm_bf16 + aql_geglu_bwd).                                                          */
int aql_gemm_bf16_geglu_bwd(const bf16_t* A, long lda, const bf16_t* B, long ldb, long M, int F, int K, const bf16_t* A2,
                            long lda2, const bf16_t* B2, long ldb2, int K2, const bf16_t* H, long ldh, bf16_t* DH, long lddh,
                            float* ws, size_t ws_bytes, aql_stream_t stream);
/* ng <= 3 rank-32 LoRA linears whose results are SUMMED into one output, in one launch:
 *   Y[M,N] = sum_g ( X_g.W_g^T + ((X_g.Adown_g^T) * S[m / rps]).Bup_g^T ) + residual;   T_g, Ts_g [M,32] are written per group.
 * The backward-data pass of the q | k | v projections of a self-attention (scripts/lib/original_unet.py:688-704 through
 * utils/lora_modules.py:56-62): X_g = dQ | dK | dV, W_g = W_g^T, Adown_g = Bup_g^T, Bup_g = A_g^T, T_g / Ts_g = dTs_g / dT_g.
 * X, ldx, W, ldw, K, Adown, Bup, T, Ts are HOST arrays of ng entries.  Returns 100 when (M, N) does not fit one chip-wide
 * round of a wave-specialised tile (caller: chained aql_lora_gemm_fused launches).                                          */
int aql_lora_gemm_fused_kgroups(int ng, const void* const* X, const long* ldx, const void* const* W, const long* ldw,
                                const int* K, const void* const* Adown, const void* const* Bup, long M, int N, const bf16_t* S,
                                int rows_per_sample, const bf16_t* residual, long ldr, bf16_t* Y, long ldy, void* const* T,
                                void* const* Ts, aql_stream_t stream);
/* n <= 32 rank-32 "down" products with a common row count in one launch: T[i] = X[i].A[i]^T, Ts[i] = T[i] * S[m / rps]
 * (X[i] [M,K[i]] dense, A[i] [32,K[i]]; X, A, K are HOST arrays; T, Ts [n][M][32]).  The backward of the grouped text-state
 * k|v projections: dTs = dY.Bup, dT = dTs * S (utils/lora_modules.py:13-17 transposed) for all 32 sites at once.        */
int aql_lora_down_grouped(int n, const bf16_t* const* X, const bf16_t* const* A, const int* K, long M, const bf16_t* S,
                          int rows_per_sample, bf16_t* T, bf16_t* Ts, aql_stream_t stream);
/* Round 6: aql_gemm_bf16_ex with COLUMN GROUPS -- output columns [g grp_n, (g + 1) grp_n) read A at a column offset of g grp_a
 * elements and A2 at g grp_a2; B / B2 rows are the output columns.  Block-diagonal products as one launch: at LoRA rank 320
 * (BASELINE config 3, train/README.md:34-48) the q | k | v projections of a self-attention (utils/lora_modules.py:9-26, 56-62 on
 * three hosts that read the same tokens; scripts/lib/original_unet.py:688-704) run
 *   forward    [T_q | T_k | T_v] = X.[A_q; A_k; A_v]^T, Ts = T * S          (aql_lora_down with r = 3 x 320 and S repeated)
 *              [q | k | v]       = X.[Wq; Wk; Wv]^T + Ts_g.Bup_g^T          (grp_n = C, grp_a2 = r)
 *   backward   [dTs_q | dTs_k | dTs_v] = dY_g.Bup_g, dT = dTs * S           (grp_n = r, grp_a = C; second output C2 = C * rowscale)
 *              dX = [dQ | dK | dV].[Wq | Wk | Wv] + [dT_q | dT_k | dT_v].[A_q; A_k; A_v]     (aql_gemm_bf16_ex, K = 3C and 3r)
 * four launches where twelve ran.  grp_n: a multiple of 320 that divides N; N % 160 == 0.  C2 (optional) = C * rowscale[m / rps][n]. */
int aql_gemm_bf16_grouped(const bf16_t* A, long lda, const bf16_t* B, long ldb, long M, int N, int K, const bf16_t* A2, long lda2,
                          const bf16_t* B2, long ldb2, int K2, int grp_n, int grp_a, int grp_a2, const bf16_t* bias,
                          const bf16_t* residual, long ldr, bf16_t* C, long ldc, bf16_t* C2, long ldc2, const bf16_t* rowscale,
                          int rows_per_sample, long lora_row0, float* ws, size_t ws_bytes, aql_stream_t stream);
int aql_conv3x3_bwd_data(const bf16_t* dY, int B, int Hin, int Win, int Cin, const bf16_t* Wt, int Cout, int stride,
                         bf16_t* dX, float* ws, size_t ws_bytes, aql_stream_t stream);
/* Round 6: a split-K convolution whose finalize launch is left to the GroupNorm behind it.  ResnetBlock2D runs
 * norm1 -> conv1 -> norm2 -> conv2 (scripts/lib/original_unet.py:423-453); at the 8x8 / 16x16 / 32x32 levels (and at every level of the
 * CFG batch-2 sampling forward, evaluation/utils_eval.py:107-126) the 3x3 convolutions split K over workgroups and a 5-6 us launch
 * summed the fp32 partial slabs.  The _defer entry points skip that launch and report the split count: *splits_out = 1 means the
 * output is complete; s > 1 means ws holds slabs [s][B*Hout*Wout][Cout] (forward; Y untouched) or [s][B*Hin*Win][Cin] (backward
 * data; dX untouched), to be finished by aql_groupnorm_silu_fwd_slabs / _bwd_slabs (one launch, bit-identical to finalize +
 * aql_groupnorm_silu_fwd / _bwd; they return 100 for maps the one-launch GroupNorm does not take) or by aql_splitk_finalize
 * (C = bf16(sum_z slabs[z] + bias) + rowbias[m / rows_per_sample] + residual, the finalize launch on its own).                   */
int aql_conv3x3_fwd_defer(const bf16_t* X, int B, int Hin, int Win, int Cin, const bf16_t* Wk, const bf16_t* bias, int Cout,
                          int stride, int upsample, const bf16_t* rowbias, long rowbias_ld, const bf16_t* residual, bf16_t* Y,
                          float* ws, size_t ws_bytes, int* splits_out, aql_stream_t stream);
int aql_conv3x3_bwd_data_defer(const bf16_t* dY, int B, int Hin, int Win, int Cin, const bf16_t* Wt, int Cout, int stride,
                               bf16_t* dX, float* ws, size_t ws_bytes, int* splits_out, aql_stream_t stream);
int aql_splitk_finalize(const float* slabs, int splits, long M, int N, const bf16_t* bias, const bf16_t* rowbias, long rowbias_ld,
                        int rows_per_sample, const bf16_t* residual, long ldr, bf16_t* C, long ldc, aql_stream_t stream);
int aql_groupnorm_silu_fwd_slabs(const float* slabs, int splits, const bf16_t* bias, const bf16_t* rowbias, long rowbias_ld,
                                 const bf16_t* residual, bf16_t* xout, int B, int HW, int C, const bf16_t* gamma,
                                 const bf16_t* beta, float eps, int silu, bf16_t* y, float* stats, aql_stream_t stream);
int aql_groupnorm_silu_bwd_slabs(const bf16_t* x, const float* slabs, int splits, int B, int HW, int C, const bf16_t* gamma,
                                 const bf16_t* beta, int silu, const float* stats, const bf16_t* dres, bf16_t* dx,
                                 aql_stream_t stream);

/* C[P,Q] += alpha * U[M,P]^T.V[M,Q] (fp32, atomics over M splits): LoRA weight gradients d(up), d(down) that
 * autograd derives from lora_modules.py:13-19.                                                                   */
int aql_gemm_tn_f32(const bf16_t* U, long ldu, const bf16_t* V, long ldv, long M, int P, int Q, float alpha, float* C,
                    long ldc, aql_stream_t stream);

/* Same contract for wide problems (LoRA rank > 32): 128x128 tiles, both token-major operands staged as they lie in
 * memory and gathered with LDS transpose reads -- no transposed copies in HBM (csrc/aql_gemm_tntr.hip).            */
int aql_gemm_tn_tr_f32(const bf16_t* U, long ldu, const bf16_t* V, long ldv, long M, int P, int Q, float alpha,
                       float* C, long ldc, aql_stream_t stream);
/* Grouped form of the above (96-byte host descriptors, same protocol as aql_tn_desc_fill / aql_gemm_tn_grouped_range):
 * one launch for all wide weight gradients of a backward pass or of one exchange bucket.                          */
int aql_tntr_desc_fill(void* host_desc, const bf16_t* U, long ldu, const bf16_t* V, long ldv, long M, int P, int Q,
                       float alpha, float* C, long ldc, int first_block);
int aql_gemm_tn_tr_grouped(const void* dev_descs, int first, int n, int block_base, int n_blocks, aql_stream_t stream);
/* Round 6: the same grouped launch on 128 x 160 tiles, for problems with a side of 320 or 960 -- the rank-320 weight gradients of
 * BASELINE config 3 (dB [C, 320], dA [320, K]: on 128-wide tiles 320 pads to 384 and every sixth MFMA multiplies zeros).
 * aql_tntr160_desc_fill returns 0 for a problem without such a side (the caller then tries aql_tntr_desc_fill); descriptors of the two
 * forms live in separate tables (ops.DeferredDW kinds "x" and "w").                                                              */
int aql_tntr160_desc_fill(void* host_desc, const bf16_t* U, long ldu, const bf16_t* V, long ldv, long M, int P, int Q, float alpha,
                          float* C, long ldc, int first_block);
int aql_gemm_tn_tr160_grouped(const void* dev_descs, int first, int n, int block_base, int n_blocks, aql_stream_t stream);

/* Grouped form: ONE launch for every LoRA weight gradient of a backward pass (all problems have a rank <= 32 side).
 * aql_tn_desc_fill writes an 80-byte descriptor into HOST memory and returns the workgroups it needs (0 = shape not
 * groupable); the caller copies the table to the device and passes the running block prefix as first_block.       */
int aql_tn_desc_fill(void* host_desc, const bf16_t* U, long ldu, const bf16_t* V, long ldv, long M, int P, int Q,
                     float alpha, float* C, long ldc, int first_block);
int aql_gemm_tn_grouped(const void* dev_descs, int n, int total_blocks, aql_stream_t stream);
/* Descriptors [first, first+n) of the same table = one gradient bucket of the data-parallel exchange (DDP's bucketed
 * all-reduce under accelerator.backward, ppft_train.py:1058); block_base = first_block of descriptor `first`.    */
int aql_gemm_tn_grouped_range(const void* dev_descs, int first, int n, int block_base, int n_blocks,
                              aql_stream_t stream);

/* ---- normalisation (csrc/aql_norm.hip) ---- torch.nn.GroupNorm(32,C,eps)+SiLU, torch.nn.LayerNorm(C) as used at
 * scripts/lib/original_unet.py:423,429,444-453,826 and :779-783.  stats: [B,32,2] / [M,2] fp32 (mean, rstd).     */
long aql_groupnorm_scratch_floats(int B, int HW);
int aql_groupnorm_silu_fwd(const bf16_t* x, int B, int HW, int C, const bf16_t* gamma, const bf16_t* beta, float eps,
                           int silu, bf16_t* y, float* stats, float* scratch, aql_stream_t stream);
/* dres (optional): gradient of the residual / shortcut branch that consumed the same x; added into dx in the same pass */
int aql_groupnorm_silu_bwd(const bf16_t* x, const bf16_t* dy, int B, int HW, int C, const bf16_t* gamma,
                           const bf16_t* beta, int silu, const float* stats, const bf16_t* dres, bf16_t* dx,
                           float* scratch, aql_stream_t stream);
int aql_layernorm_fwd(const bf16_t* x, long M, int C, const bf16_t* gamma, const bf16_t* beta, float eps, bf16_t* y,
                      float* stats, aql_stream_t stream);
int aql_layernorm_bwd(const bf16_t* x, const bf16_t* dy, long M, int C, const bf16_t* gamma, const float* stats,
                      const bf16_t* dres, bf16_t* dx, aql_stream_t stream);
/* Row-resident CHAIN of the transformer block at the 320-channel level (round 5): up to 4 rank-32 LoRA linears 320 -> 320 whose
 * inputs and outputs never leave the CU between two of them -- one workgroup owns 128 token rows, the 128 x 320 activation tile
 * stays in LDS as the A panel of the next linear, weights stream through an LDS-DMA ring.  Replaces, bit for bit, the launch
 * sequences   attn.to_out (+ residual) -> LayerNorm -> next projection(s)   of BasicTransformerBlock.forward
 * (scripts/lib/original_unet.py:786-806 with utils/lora_modules.py:9-26, 56-62 on every linear) and
 * proj_in -> norm1 -> to_q | to_k | to_v (original_unet.py:856-861, 786-790):
 *   stage g:  Y = X.W_g^T + ((X.Adown_g^T) * S[m / rps]).Bup_g^T + bias_g          X = the tile in LDS ([M, 320] input for g = 0)
 *     keep_g = 1:  X <- bf16(Y) + res_g (bf16 add);  out_g <- X (if given);  ln_g = 1:  X <- LayerNorm(X; gamma_g, beta_g, eps_g),
 *                  stats_g[m] = (mean, rstd),  nout_g <- X for rows >= nout_row0_g
 *     keep_g = 0:  out_g <- bf16(Y * oscale_g)                                   (X unchanged: q | k | v read the same tile;
 *                  oscale (host array or null = all 1): scale log2(e) on attn1.to_q, whose consumer is then aql_sdpa_fwd_qpre)
 * Every per-stage argument is a HOST array of nstage entries (null entries where a stage has no such operand; Adown_g = null:
 * no LoRA on that linear).  M, rows_per_sample, lora_row0, nout_row0: multiples of 64 (128-row tiles run when all are multiples
 * of 128 and fill the chip, 64-row tiles otherwise; rows below lora_row0 -- the clean half of a twin batch,
 * ppft_train.py:1026-1029 -- carry no LoRA term and write no T / Ts); at most one LayerNorm per chain.                         */
int aql_lora_chain_fwd(const bf16_t* X, long ldx, long M, int rows_per_sample, long lora_row0, const bf16_t* S, int nstage,
                       const void* const* W, const long* ldw, const void* const* bias, const void* const* Adown,
                       const void* const* Bup, void* const* T, void* const* Ts, const void* const* res, const long* ldr,
                       void* const* out, const long* ldo, const int* keep, const int* ln, const void* const* gamma,
                       const void* const* beta, const float* eps, void* const* stats, void* const* nout, const long* ldn,
                       const long* nout_row0, const float* oscale, aql_stream_t stream);
/* The same chain at LoRA rank 320 (BASELINE config 3, train/README.md:34-48): Adown_g [320][320] (rank x K), Bup_g [320][320]
 * (N x rank), T_g / Ts_g [M][320], S [M / rps][320].  One workgroup owns 64 token rows; per LoRA linear the down product
 * T = X.Adown^T runs as a pass of its own, (T, Ts = T * S) go to HBM for backward and Ts stays in LDS as the second A panel of the
 * main pass X.W^T + Ts.Bup^T.  Bit for bit aql_lora_down + the two-K-segment aql_gemm_bf16.  M, rows_per_sample, lora_row0: % 64.  */
int aql_lora_chain_fwd_r320(const bf16_t* X, long ldx, long M, int rows_per_sample, long lora_row0, const bf16_t* S, int nstage,
                            const void* const* W, const long* ldw, const void* const* bias, const void* const* Adown,
                            const void* const* Bup, void* const* T, void* const* Ts, const void* const* res, const long* ldr,
                            void* const* out, const long* ldo, const int* keep, const int* ln, const void* const* gamma,
                            const void* const* beta, const float* eps, void* const* stats, void* const* nout, const long* ldn,
                            const long* nout_row0, const float* oscale, aql_stream_t stream);
/* The mirrored BACKWARD chains (round 5): the backward-data passes of up to 4 of those linears with the LayerNorm backward between
 * them, one launch, 64-row tiles (the backward pass runs on the watermarked half of the batch only).  Per linear the operands of
 * aql_lora_gemm_fused's backward-data form: Wt = W^T [320][ldw], BupT = Bup^T [32][320], AT = A^T [320][32]; dTs, dT [M][32] out.
 *   tile <- dY rows;   [ln_x[0] given: tile <- LayerNormBackward(tile; ln_x[0], ln_stats[0], ln_gamma[0]) + ln_dres[0];  ln_out[0] <- tile]
 *   stage g:  dXg = tile.Wt_g^T + ((tile.BupT_g^T) * S[m / rps]).AT_g^T
 *     keep_g = 1:  tile <- bf16(dXg);  ln_x[g + 1] given: tile <- LayerNormBackward(tile; ...[g + 1]) + ln_dres[g + 1];  ln_out[g + 1] <- tile
 *     keep_g = 0:  dX_g <- bf16(dXg)                                                      (tile unchanged)
 * LayerNormBackward is aql_layernorm_bwd's arithmetic per row (x = the LayerNorm's saved input, stats = its saved (mean, rstd), the
 * residual branch's gradient ln_dres added last).  Replaces, bit for bit, the backward launches of BasicTransformerBlock
 * (scripts/lib/original_unet.py:786-806 under autograd):  norm3 backward -> attn2.to_out backward;  attn2.to_q backward -> norm2
 * backward -> attn1.to_out backward;  norm1 backward -> proj_in backward.  The ln_* arrays have nstage + 1 entries (entry 0: the pass
 * on the chain input), every other per-stage argument nstage entries; all are HOST arrays.  At most one LayerNorm per chain.      */
int aql_lora_chain_bwd(const bf16_t* dY, long lddy, long M, int rows_per_sample, const bf16_t* S, int nstage,
                       const void* const* Wt, const long* ldw, const void* const* BupT, const void* const* AT,
                       void* const* dTs, void* const* dT, void* const* dX, const long* lddx, const int* keep,
                       const void* const* ln_x, const long* ld_lnx, const void* const* ln_stats, const void* const* ln_gamma,
                       const void* const* ln_dres, const long* ld_dres, void* const* ln_out, const long* ld_lnout,
                       aql_stream_t stream);

/* ---- attention (csrc/aql_attn.hip) ---- F.scaled_dot_product_attention via diffusers AttnProcessor2_0 / twin
 * original_unet.py:688-704.  q/k/v/o: [B,N,H*d] with row strides ld*; lse,delta: [B,H,Nq] fp32.                   */
int aql_sdpa_fwd(const bf16_t* q, long ldq, const bf16_t* k, long ldk, const bf16_t* v, long ldv, int B, int H, int Nq,
                 int Nk, int d, float scale, bf16_t* o, long ldo, float* lse, aql_stream_t stream);
int aql_sdpa_bwd(const bf16_t* q, long ldq, const bf16_t* k, long ldk, const bf16_t* v, long ldv, const bf16_t* o,
                 const bf16_t* dout, long ldo, const float* lse, float* delta, int B, int H, int Nq, int Nk, int d,
                 float scale, bf16_t* dq, bf16_t* dk, bf16_t* dv, float* ws, size_t ws_bytes, aql_stream_t stream);
/* The same attention with q PRE-MULTIPLIED by scale * log2(e) in the epilogue of the launch that produced it (one bf16 rounding of
 * q c instead of q: aql_lora_chain_fwd's `oscale` on attn1.to_q, scripts/lib/original_unet.py:688-704): the forward loop then carries
 * its softmax shift inside the S-product (no multiply-add per score) at the precision of aql_sdpa_fwd.  `scale` is still passed
 * (dQ is the gradient of the UNSCALED q: what the producing linear's backward expects); lse is the same natural-log quantity.     */
int aql_sdpa_fwd_qpre(const bf16_t* q, long ldq, const bf16_t* k, long ldk, const bf16_t* v, long ldv, int B, int H, int Nq,
                      int Nk, int d, float scale, bf16_t* o, long ldo, float* lse, aql_stream_t stream);
int aql_sdpa_bwd_qpre(const bf16_t* q, long ldq, const bf16_t* k, long ldk, const bf16_t* v, long ldv, const bf16_t* o,
                      const bf16_t* dout, long ldo, const float* lse, float* delta, int B, int H, int Nq, int Nk, int d,
                      float scale, bf16_t* dq, bf16_t* dk, bf16_t* dv, float* ws, size_t ws_bytes, aql_stream_t stream);

/* Round 6: either backward (qpre = 0 / 1) with the gradients at row strides of ldg_q (dq) / ldg_kv (dk, dv) elements (0 = dense):
 * dq | dk | dv as column blocks of one [B][N][3 H d] buffer, or dk | dv of a text-state attention as column blocks of one
 * [B][Nk][2 H d] buffer -- read in place by the grouped backward of the projections (aql_gemm_bf16_grouped).                        */
int aql_sdpa_bwd_ex(int qpre, const bf16_t* q, long ldq, const bf16_t* k, long ldk, const bf16_t* v, long ldv, const bf16_t* o,
                    const bf16_t* dout, long ldo, const float* lse, float* delta, int B, int H, int Nq, int Nk, int d, float scale,
                    bf16_t* dq, bf16_t* dk, bf16_t* dv, long ldg_q, long ldg_kv, float* ws, size_t ws_bytes, aql_stream_t stream);
/* ws (optional, caller-owned, per stream): fp32 scratch for split-Q partials of dK/dV when Nk is too short to fill the
 * chip (cross-attention, Nk = 77); 2*splits*B*H*Nk*d floats are used, NULL disables the split.                    */

/* ---- elementwise / small modules (csrc/aql_elem.hip) ---------------------------------------------------------- */
/* GEGLU  original_unet.py:727-729 : out[M,F] = in[:, :F] * gelu(in[:, F:])                                         */
int aql_geglu_fwd(const bf16_t* in, long M, int F, bf16_t* out, aql_stream_t stream);
int aql_geglu_bwd(const bf16_t* in, const bf16_t* dy, long M, int F, bf16_t* din, aql_stream_t stream);
/* backward of F.interpolate(scale_factor=2, "nearest") (original_unet.py:1076): 2x2 block sum                      */
int aql_upsample2x_bwd(const bf16_t* du, int B, int H, int W, int C, bf16_t* dx, aql_stream_t stream);
/* Skip-connection concat of the up blocks (torch.cat([h, skip], dim=1), scripts/lib/original_unet.py:1133,1224) on channels-last
 * maps: cat[p][0:Ca] = a[p], cat[p][Ca:Ca+Cb] = b[p] for npix = B*H*W pixels in one launch, and its backward (the two dense
 * gradient slices in one launch).  Ca, Cb multiples of 8.                                                               */
int aql_cat_channels(const bf16_t* a, const bf16_t* b, long npix, int Ca, int Cb, bf16_t* cat, aql_stream_t stream);
int aql_split_channels(const bf16_t* cat, long npix, int Ca, int Cb, bf16_t* a, bf16_t* b, aql_stream_t stream);
/* P[m,:] = softmax(scale * S[m,:]), S fp32 -> P bf16: the VAE mid-block's single-head 512-wide attention
 * (AutoencoderKL, ppft_train.py:993), whose scores come from aql_gemm_nt_f32_accum                                   */
int aql_softmax_rows(const float* S, long lds, long M, int N, float scale, bf16_t* P, long ldp, aql_stream_t stream);
/* its backward (stage 1 trains the SecretEncoder THROUGH vae.decode, latent_wm_pretrain.py:180-181):
 * dS[m,:] = scale * P[m,:] * (dP[m,:] - sum_j P[m,j] dP[m,j])                                                      */
int aql_softmax_rows_bwd(const bf16_t* P, const bf16_t* dP, long ld, long M, int N, float scale, bf16_t* dS,
                         aql_stream_t stream);
/* ---- CLIP text encoder pieces (transformers CLIPTextModel; text_encoder(input_ids)[0] at train/ppft_train.py:1014-1019) */
/* CLIPMLP activation quick_gelu(x) = x * sigmoid(1.702 x); n % 8 == 0                                               */
int aql_quick_gelu(const bf16_t* x, long n, bf16_t* y, aql_stream_t stream);
/* causal self-attention for short sequences (N <= 128, d <= 128): q/k/v/o [B,N,H*d], softmax(scale q k^T + causal) v */
int aql_causal_attn_small(const bf16_t* q, const bf16_t* k, const bf16_t* v, long ld, int B, int H, int N, int d,
                          float scale, bf16_t* o, long ldo, aql_stream_t stream);
/* DDPMScheduler.add_noise on x0 and x0+wm with shared noise/timesteps  train/ppft_train.py:1010-1011              */
int aql_add_noise(const float* x0, const float* wm, const float* eps, const long* t, const float* alphas_cumprod, int B,
                  int per_sample, bf16_t* noisy, bf16_t* noisy_wm, aql_stream_t stream);
/* F.mse_loss(pred.float(), target.float()) and its gradient  train/ppft_train.py:1051                              */
int aql_mse_fwd_bwd(const bf16_t* pred, const bf16_t* target, long n, float* loss, bf16_t* dpred, aql_stream_t stream);
/* MapperNet.forward / its weight gradient  utils/models.py:110-115                                                 */
/* Everything between the batch and the first GEMM of the twin (clean | watermarked) PPFT forward in one launch:
 * noisy latents of both passes (train/ppft_train.py:1010-1011, utils/cschedulers.py:15: aql_add_noise's arithmetic) written
 * channels-last at conv_in's packed width, the text states twice, the sinusoidal timestep embedding
 * (scripts/lib/original_unet.py:323-361, flip_sin_to_cos), MapperNet (utils/models.py:110-115) as S32 / S16 = [0 | S], and
 * the zeroed dS accumulator.  Layouts in csrc/aql_elem.hip.                                                              */
int aql_ppft_prologue(const float* z, const float* wm, const float* eps, const long* t, const float* acp, const float* msg,
                      const float* E, const float* freq, const void* ctx, int ctx_f32, int B, int HW, int bits, int r,
                      int half, long ctx_per_sample, bf16_t* x2, bf16_t* ctx2, bf16_t* temb, float* S32, bf16_t* S16,
                      float* ds, aql_stream_t stream);
int aql_mapper_fwd(const float* msg, const float* E, int nb, int bits, int r, float* S32, bf16_t* S16,
                   aql_stream_t stream);
int aql_mapper_bwd(const float* msg, const float* dS, int nb, int bits, int r, float* dE, aql_stream_t stream);
/* SecretEncoder.encode  utils/models.py:57-64,70-72 (out = conv(...) * out_scale, NCHW fp32)                        */
int aql_secret_encoder_fwd(const float* msg, const float* lin_w, const float* lin_b, const float* conv_w,
                           const float* conv_b, int nb, int bits, int base_res, int res, float out_scale,
                           float* hidden_scratch, float* out, aql_stream_t stream);
/* fp32 LoRA master weight [rows,cols] -> bf16 copy and bf16 transposed copy (autocast's casts, lora_modules.py:10-13) */
int aql_cast_transpose(const float* w, int rows, int cols, bf16_t* out, bf16_t* outT, aql_stream_t stream);
/* all LoRA tensors in one launch: desc = device array of {const float* w; bf16_t* out; bf16_t* outT; int rows, cols,
 * first_tile, pad} (32 bytes each, 32x32 tiles numbered consecutively)                                              */
int aql_cast_transpose_batched(const void* desc, int n, int total_tiles, aql_stream_t stream);
/* dS[b,j] += sum_{m in sample b} dTs[m,j]*T[m,j]: gradient of the diagonal (autograd of diag_embed, :16-17)         */
int aql_lora_ds(const bf16_t* dTs, const bf16_t* T, int nb, int rows_per_sample, int r, float* dS, aql_stream_t stream);
/* Weight-side form of the LoRA linear (aql_gemm_bf16_sw): with P[b][n][j] = sum_k (dY_b^T X_b)[n][k] A[j][k] (bf16, [B][N][r]),
 * dBup[n][j] += sum_b P[b][n][j] S[b][j]  and  dS[b][j] += sum_n Bup[n][j] P[b][n][j]  (the gradients of up.weight and of the
 * diagonal in utils/lora_modules.py:13-19).  dS must tolerate fp32 atomics (zeroed by the caller once per step).                 */
int aql_wside_reduce(const bf16_t* P, const bf16_t* S, const bf16_t* Bup, int B, int N, int r, float* dBup, long lddb, float* dS,
                     aql_stream_t stream);
/* grouped form (48-byte host descriptors, same protocol as aql_tn_desc_fill)                                         */
int aql_ds_desc_fill(void* host_desc, const bf16_t* dTs, const bf16_t* T, int nb, int rows_per_sample, int r, float* dS,
                     int first_block);
/* ... with dTs at a row stride of ld_dts elements (0 = dense): a column block of a stacked [M, G r] product (round 6)          */
int aql_ds_desc_fill_ld(void* host_desc, const bf16_t* dTs, long ld_dts, const bf16_t* T, int nb, int rows_per_sample, int r,
                        float* dS, int first_block);
int aql_lora_ds_grouped(const void* dev_descs, int n, int total_blocks, aql_stream_t stream);
/* clip_grad_norm_ + torch.optim.AdamW on flat fp32 buffers  train/ppft_train.py:1059-1066, 779-787                 */
int aql_sumsq_f32(const float* g, long n, float* out, aql_stream_t stream);
int aql_clipnorm_adamw(float* p, const float* g, float* m, float* v, long n, const float* sumsq, float max_norm,
                       const float* lr, float beta1, float beta2, float eps, float wd, const int* step,
                       aql_stream_t stream);

/* ---- distortion layers (csrc/aql_jpeg.hip) ---- JpegCompression.forward  utils/noise_layers/jpeg_compression.py:127-162
 * (RGB->YUV, 8x8 DCT, zig-zag keep-mask 25/9/9, IDCT, YUV->RGB; NCHW fp32).  backward=1 applies the transposed map.   */
int aql_jpeg_mask(const float* x, float* y, int B, int H, int W, int keep_y, int keep_u, int keep_v, int backward,
                  aql_stream_t stream);

/* DDIM step (eta 0, epsilon prediction) with classifier-free guidance, in place on the fp32 latents: the sampler maths of
 * evaluation/utils_eval.py:83-126 (`--sampler ddim`) / diffusers DDIMScheduler.step.  coef = {sqrt(a_t), sqrt(1-a_t),
 * sqrt(a_prev), sqrt(1-a_prev)} on the device.                                                                          */
int aql_ddim_step(float* x, const bf16_t* eps_uncond, const bf16_t* eps_cond, float guidance, const float* coef, long n,
                  aql_stream_t stream);
/* DPM-Solver++ 2M step (DPMSolverMultistepScheduler as set at train/rob_enhance_finetune.py:993, 20 steps at :1012):
 * eps = CFG(eps_u, eps_c); x0 = (x - sigma_t eps) / alpha_t; x <- a x + b x0 + c x0_prev; x0_prev <- x0;
 * coef (device) = {alpha_t, sigma_t, a, b, c}                                                                         */
int aql_dpmpp2m_step(float* x, const bf16_t* eps_u, const bf16_t* eps_c, float guidance, float* x0_prev,
                     const float* coef, long n, aql_stream_t stream);
/* One phase of a captured sampling loop for the OTHER schedulers of evaluation/utils_eval.py:83-102 (Euler, Heun, KDPM2, KDPM2-
 * ancestral, LMS, PLMS, DPM-Solver++ single-step, UniPC): every update between two U-Net calls is a linear combination of the state x,
 * one auxiliary state, four history entries, caller-supplied noise and the guided prediction e = eps_u + g (eps_c - eps_u); coef
 * (12 floats) and flag (5 ints) live in DEVICE memory so that one hipGraph replays for every phase (layout: csrc/aql_elem.hip).
 * Also writes the next model input uin [2n] = next_in_scale * (x | aux), both halves of the guidance batch.                       */
int aql_sampler_step(float* x, float* aux, float* hist, const float* noise, const bf16_t* eps_u, const bf16_t* eps_c, float* uin,
                     const float* coef, const int* flag, long n, aql_stream_t stream);

/* csrc/aql_distort.hip: deterministic image maps of noises.py:34-85 / noiser.py:46-71 (random parameters are drawn by the
 * host like the reference does); NCHW fp32, BC = batch*channels; backward=1 applies the adjoint.                        */
int aql_crop_resize_bilinear(const float* src, float* dst, int BC, int H, int W, int top, int left, int ch, int cw,
                             int oh, int ow, int backward, aql_stream_t stream);
int aql_gauss_blur(const float* src, float* dst, float* tmp, int BC, int H, int W, int k, const float* taps, int backward,
                   aql_stream_t stream);
/* kornia RandomGaussianBlur((ky, kx), sigma) as the reference calls it (noises.py:68 (3,9); noiser.py:63 (3,5);
 * utils_eval.py:280 (3,3)): anisotropic separable kernel, reflect border, taps shared or one row per sample            */
int aql_gauss_blur2(const float* src, float* dst, float* tmp, int B, int C, int H, int W, int kx, int ky,
                    const float* taps_x, const float* taps_y, int per_sample, int backward, aql_stream_t stream);
int aql_add_gauss_noise(const float* x, const float* noise, float std, int clamp01, long n, float* y,
                        aql_stream_t stream);
/* kornia ColorJiggle as called at noises.py:97-103, noiser.py:52-57, utils_eval.py:271-276 (kornia 0.6.12, recalled):
 * x,y [B,3,H,W] in [0,1]; factors device [B][4] = {brightness-1, contrast, saturation, hue*2pi}; order device int[4]
 * (permutation of 0 brightness, 1 contrast, 2 saturation, 3 hue).  dy == NULL: y = f(x); else y = J(x)^T dy.          */
int aql_color_jiggle(const float* x, const float* dy, float* y, int B, int H, int W, const float* factors,
                     const int* order, aql_stream_t stream);
/* kornia RandomRotation -> rotate (noises.py:20-31, utils_eval.py:292): bilinear, zeros padding, align_corners=True,
 * about the image centre; angle_deg device [B], anti-clockwise.  backward=1: src = dy, dst = dx.                       */
int aql_rotate_bilinear(const float* src, float* dst, int B, int C, int H, int W, const float* angle_deg, int backward,
                        aql_stream_t stream);
/* kornia RandomSharpness -> sharpness (noises.py:106-119, utils_eval.py:294): factor device [B].  dy == NULL:
 * dst = sharpness(x); else dst = dx and tmp is B*C*H*W floats of scratch.                                              */
int aql_sharpness(const float* x, const float* dy, float* dst, float* tmp, int B, int C, int H, int W,
                  const float* factor, aql_stream_t stream);

/* ---- SecretDecoder inference (csrc/aql_decoder.hip) ---- utils/models.py:91-96 (torchvision efficientnet_b1, eval mode,
 * BatchNorm folded by the host), fp32 NHWC.                                                                           */
int aql_resize_bilinear_nhwc(const float* x_nchw, int B, int C, int H, int W, int Ho, int Wo, float* y_nhwc,
                             aql_stream_t stream);
int aql_stem_conv3x3s2_silu(const float* x, const float* w, const float* bias, int B, int H, int W, int Cout, float* y,
                            aql_stream_t stream);
int aql_dwconv_silu(const float* x, const float* w, const float* bias, int B, int H, int W, int C, int k, int stride,
                    float* y, aql_stream_t stream);
int aql_avgpool_nhwc(const float* x, int B, int HW, int C, float* out, aql_stream_t stream);
int aql_se_gate(const float* pool, const float* w1, const float* b1, const float* w2, const float* b2, int B, int C,
                int Cs, float* gate, aql_stream_t stream);
int aql_pwconv_f32(const float* x, const float* w, const float* bias, const float* gate, int rows_per_sample,
                   const float* residual, long M, int N, int K, int act, float* y, aql_stream_t stream);
/* the squeeze-excite pool as S pixel slabs (a chip-wide pass instead of one workgroup per sample and 64 channels): part [B][S][C]
 * raw sums, added in slab order and divided by HW inside aql_se_gate_slabs (same reference lines as aql_avgpool_nhwc / aql_se_gate) */
int aql_avgpool_nhwc_slabs(const float* x, int B, int HW, int C, int S, float* part, aql_stream_t stream);
int aql_se_gate_slabs(const float* part, int S, int HW, const float* w1, const float* b1, const float* w2, const float* b2, int B,
                      int C, int Cs, float* gate, aql_stream_t stream);

/* wide-rank (r > 32) LoRA weight gradients (ppft_train.py:1058 backward through lora_modules.py:13-19): both operands
 * are transposed once (aql_transpose_bf16: dst[cols][rows] = src[rows][cols]^T) and C[M,N] += alpha * A[M,K].B[N,K]^T runs
 * on the pipelined NT kernels with split-K slabs in ws                                                                    */
int aql_transpose_bf16(const bf16_t* src, long rows, int cols, long ld, bf16_t* dst, aql_stream_t stream);
int aql_gemm_nt_f32_accum(const bf16_t* A, long lda, const bf16_t* B, long ldb, long M, int N, long K, float alpha,
                          float* C, long ldc, float* ws, size_t ws_bytes, aql_stream_t stream);

/* ---- SecretDecoder training (csrc/aql_decoder_train.hip) ---- torchvision efficientnet_b1 in train() mode as run by
 * train/latent_wm_pretrain.py:159-225 (sec_decoder.train(), :160) and train/rob_enhance_finetune.py (msgdecoder fwd+bwd at
 * B=16): BatchNorm with batch statistics and every layer's backward.  fp32, channels-last [B,H,W,C] == [M,C].            */
/* C[m,n] = sum_k A[m*sam + k*sak] * B[n*sbn + k*sbk] (+ bias[n]); one stride of each operand must be 1 (1x1 conv fwd /
 * bwd-data / bwd-weight are the three stride patterns)                                                                   */
int aql_gemm_f32(const float* A, long sam, long sak, const float* B, long sbn, long sbk, const float* bias, float* C,
                 long ldc, long M, int N, long K, aql_stream_t stream);
long aql_bn_scratch_floats(long M, int C);
/* nn.BatchNorm2d(eps, momentum) training forward (+ fused SiLU when act=1); mean/invstd [C] are saved for the backward   */
int aql_bn_train_fwd(const float* x, const float* gamma, const float* beta, long M, int C, float eps, float momentum,
                     int act, float* y, float* mean, float* invstd, float* run_mean, float* run_var, float* scratch,
                     aql_stream_t stream);
int aql_bn_train_bwd(const float* x, const float* dy, const float* gamma, const float* beta, const float* mean,
                     const float* invstd, long M, int C, int act, float* dx, float* dgamma, float* dbeta, float* scratch,
                     aql_stream_t stream);
/* Round 6, the first fusion of the decoder step: the last BatchNorm of an MBConv block with a skip connection (torchvision
 * MBConv.forward: result = stochastic_depth(block(x)); result += x -- utils/models.py:84-96 builds efficientnet_b1) as
 *   y = rowscale[m / rows_per_sample] * act(BN_train(x)) + res
 * in the BatchNorm's apply pass (rowscale [M / rows_per_sample]: the per-sample survival factor, null = 1; res: the skip, null = none),
 * and its backward for x (dy = d(y); d(res) = dy is the caller's): the per-sample scale multiplies dy on the fly.              */
int aql_bn_train_fwd_res(const float* x, const float* gamma, const float* beta, long M, int C, float eps, float momentum, int act,
                         const float* res, const float* rowscale, long rows_per_sample, float* y, float* mean, float* invstd,
                         float* run_mean, float* run_var, float* scratch, aql_stream_t stream);
int aql_bn_train_bwd_rs(const float* x, const float* dy, const float* gamma, const float* beta, const float* mean, const float* invstd,
                        long M, int C, int act, const float* rowscale, long rows_per_sample, float* dx, float* dgamma, float* dbeta,
                        float* scratch, aql_stream_t stream);
/* depthwise k x k conv, w packed [k*k][C]: mode 0 forward, 1 backward-data (src = dy), 2 backward-weight (src = x,
 * src2 = dy, dst = dw)                                                                                                   */
int aql_dwconv_train(const float* src, const float* src2, const float* w, int B, int H, int W, int C, int k, int stride,
                     int mode, float* dst, aql_stream_t stream);
/* stem 3x3 stride-2 conv 3 -> Cout, w packed [27][Cout]; modes as above                                                  */
int aql_stem_train(const float* src, const float* src2, const float* w, int B, int H, int W, int Cout, int mode,
                   float* dst, aql_stream_t stream);
/* squeeze-excite / pooling plumbing on [B,HW,C] with per-(sample, channel) vectors [B,C]                                 */
int aql_chan_scale(const float* x, const float* g, int B, long HW, int C, float* y, aql_stream_t stream);
int aql_chan_reduce(const float* a, const float* bmul, int B, long HW, int C, float scale, float* out,
                    aql_stream_t stream);
int aql_chan_bcast(const float* g, int B, long HW, int C, float scale, int accumulate, float* dx, aql_stream_t stream);
/* kind 1 SiLU, 2 sigmoid; dy == NULL: y = act(x), else y = dy * act'(x)                                                  */
int aql_act_f32(const float* x, const float* dy, int kind, long n, float* y, aql_stream_t stream);
int aql_resize_bilinear_nhwc_bwd(const float* dy, int B, int C, int H, int W, int Ho, int Wo, float* dx,
                                 aql_stream_t stream);
/* F.binary_cross_entropy_with_logits (mean), latent_wm_pretrain.py:196; dlogits may be NULL                              */
int aql_bce_logits(const float* logits, const float* target, long n, float* loss, float* dlogits, aql_stream_t stream);

/* ---- stage 1, latent watermark pre-training (csrc/aql_stage1.hip) ---- train/latent_wm_pretrain.py:159-225             */
/* gradients of SecretEncoder.encode() (utils/models.py:70-73): dout [nb,4,res,res]; hidden = forward's SiLU output       */
int aql_secret_encoder_bwd(const float* dout, const float* msg, const float* lin_w, const float* lin_b,
                           const float* conv_w, const float* hidden, int nb, int bits, int base_res, int res,
                           float* dpre_scratch, float* dlin_w, float* dlin_b, float* dconv_w, float* dconv_b,
                           aql_stream_t stream);
/* PRVL_loss (latent_wm_pretrain.py:42-50): max over (sample, window) of the win x win box mean (padding win/2) of the
 * channel-mean |img1 - img2|; loss is a device scalar, arg the winning window (device long) for the backward              */
long aql_prvl_scratch_floats(int B, int H, int W, int win);
int aql_prvl_loss_fwd(const float* img1, const float* img2, int B, int C, int H, int W, int win, float* scratch,
                      float* loss, long* arg, aql_stream_t stream);
int aql_prvl_loss_bwd(const float* img1, const float* img2, const long* arg, const float* gout, int B, int C, int H, int W,
                      int win, float* d1, float* d2, aql_stream_t stream);

/* ---- csrc/aql_lpips.hip: LPIPS(VGG16) around the 3x3 conv kernels -- stage 1's perceptual loss
 * (`lpips.LPIPS(net='vgg')`, train/latent_wm_pretrain.py:111; `loss_fn_vgg(clean_image, watermarked_image)`, :182).
 * aql_lpips_scale: ScalingLayer, fp32 NCHW [B,3,H,W] -> bf16 NHWC [B,H,W,8] (channels 3..7 zero) and its adjoint;
 * aql_relu_bf16 / aql_maxpool2x2_nhwc: the VGG activations and pools with their backward;
 * aql_lpips_layer: out[b] += mean_hw sum_c w_c (f0/(|f0|+1e-10) - f1/(|f1|+1e-10))^2 (normalize_tensor + lin head + spatial
 * average of one feature tap; `out` accumulates the five taps), aql_lpips_layer_bwd: its gradient w.r.t. f1.            */
int aql_lpips_scale(const float* x, int B, int H, int W, bf16_t* y, aql_stream_t stream);
int aql_lpips_scale_bwd(const bf16_t* dy, int B, int H, int W, float* dx, aql_stream_t stream);
int aql_relu_bf16(const bf16_t* x, long n, bf16_t* y, aql_stream_t stream);
int aql_relu_bf16_bwd(const bf16_t* dy, const bf16_t* y, long n, bf16_t* dx, aql_stream_t stream);
int aql_maxpool2x2_nhwc(const bf16_t* x, int B, int H, int W, int C, bf16_t* y, aql_stream_t stream);
int aql_maxpool2x2_nhwc_bwd(const bf16_t* x, const bf16_t* y, const bf16_t* dy, int B, int H, int W, int C, bf16_t* dx,
                            aql_stream_t stream);
int aql_lpips_layer(const bf16_t* f0, const bf16_t* f1, const float* w, int B, long HW, int C, float* out,
                    aql_stream_t stream);
int aql_lpips_layer_bwd(const bf16_t* f0, const bf16_t* f1, const float* w, int B, long HW, int C, const float* gout,
                        bf16_t* df1, aql_stream_t stream);

/* ---- csrc/aql_comm.hip: the data-parallel exchange over RCCL / xGMI (one process per GPU) ---------------------------------
 * Replaces what accelerate's DDP wrapper does for the trainable LoRA + mapper parameters: the construction-time parameter
 * broadcast (train/ppft_train.py:905-912, accelerator.prepare) and the gradient all-reduce(mean) fired from
 * accelerator.backward (:1058); aql_comm_all_gather also serves the logged-loss gather (:1054).  Every collective is enqueued
 * on the CALLER's stream: it can sit on a forked side stream under the rest of backward and be captured into the step's
 * hipGraph.  RCCL is bound at run time (dlopen librccl.so.1: the instance PyTorch-ROCm already loaded, when there is one).
 *   aql_comm_available   1 if RCCL resolves in this process (agreed across ranks before the collective init), else 0
 *   aql_comm_unique_id   rank 0: 128 opaque bytes, handed to every rank by the launcher's side channel
 *   aql_comm_init        collective: communicator of the current HIP device  ->  *comm
 *   aql_comm_size        number of ranks (or -1): a count, not a status
 *   aql_comm_all_reduce_f32      buf <- sum | mean over ranks, in place
 *   aql_comm_reduce_scatter_f32  recv[recv_n] <- this rank's slice of the sum | mean of send[recv_n * nranks]
 *   aql_comm_all_gather          recv[rank * send_bytes ...] <- send of every rank (ZeRO-1 parameter gather, loss gather)
 *   aql_comm_broadcast           buf of `root` overwrites everyone's
 *   aql_comm_abort / aql_comm_destroy
 * aql_abi_version: AQL_ABI_VERSION of the built library; bumped when an existing entry point changes its signature
 * (round 2 inserted `lora_row0` into aql_lora_gemm_fused): a binding built against another version must refuse to load.   */
#include "aqualora_abi.h" /* AQL_ABI_VERSION */
int aql_abi_version(void);
int aql_comm_available(void);
int aql_comm_unique_id(void* id128);
int aql_comm_init(const void* id128, int nranks, int rank, void** comm);
int aql_comm_size(void* comm);
int aql_comm_all_reduce_f32(void* comm, float* buf, long n, int average, aql_stream_t stream);
int aql_comm_reduce_scatter_f32(void* comm, const float* send, float* recv, long recv_n, int average, aql_stream_t stream);
int aql_comm_all_gather(void* comm, const void* send, void* recv, long send_bytes, aql_stream_t stream);
int aql_comm_broadcast(void* comm, void* buf, long nbytes, int root, aql_stream_t stream);
int aql_comm_abort(void* comm);
int aql_comm_destroy(void* comm);

#ifdef __cplusplus
}
#endif
#endif
