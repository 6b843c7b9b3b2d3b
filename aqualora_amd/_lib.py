"""ctypes binding of the gfx950 C-ABI library (``aqualora_amd/csrc/libaqualora_hip.so``).

The library is the product: there is no CPU or PyTorch fallback behind these entry points.  If the shared
object is missing or a symbol declared in ``include/aqualora_hip.h`` cannot be resolved, importing the
compute path raises immediately.
"""
import ctypes
import os

import torch  # noqa: F401  (must be imported first so that libamdhip64.so.7 resolves to torch's HIP runtime)

_HERE = os.path.dirname(os.path.abspath(__file__))
# AQL_LIB selects another build of the same ABI (kernel A/B runs on one box, tools/ab_bench.sh)
LIB_PATH = os.environ.get("AQL_LIB") or os.path.join(_HERE, "csrc", "libaqualora_hip.so")

c_p = ctypes.c_void_p
c_i = ctypes.c_int
c_l = ctypes.c_long
c_f = ctypes.c_float
c_sz = ctypes.c_size_t

# name -> argtypes.  Mirrors include/aqualora_hip.h (tests/test_abi.py checks the two stay in sync).
SIGNATURES = {
    "aql_gemm_bf16": [c_p, c_l, c_p, c_l, c_l, c_i, c_i, c_p, c_l, c_p, c_l, c_i, c_p, c_p, c_i, c_p, c_l, c_p, c_l,
                      c_p, c_sz, c_p],
    "aql_gemm_bf16_ex": [c_p, c_l, c_p, c_l, c_l, c_i, c_i, c_p, c_l, c_p, c_l, c_i, c_p, c_p, c_i, c_p, c_l, c_p, c_l,
                         c_l, c_p, c_sz, c_p],
    "aql_lora_gemm_fused": [c_p, c_l, c_p, c_l, c_l, c_i, c_i, c_p, c_p, c_i, c_p, c_p, c_p, c_l, c_p, c_l, c_p, c_p, c_l, c_p],
    # A lda B ldb M N K | Bs sstride srows srow0 | bias residual ldr res_mod | C ldc G ldg geglu_F c_row0 | gb_h gb_ldh | ws ws_bytes stream
    "aql_gemm_bf16_sw": [c_p, c_l, c_p, c_l, c_l, c_i, c_i, c_p, c_l, c_l, c_l, c_p, c_p, c_l, c_l, c_p, c_l, c_p, c_l, c_i, c_l,
                         c_p, c_l, c_p, c_sz, c_p],
    "aql_gemm_bf16_geglu": [c_p, c_l, c_p, c_l, c_l, c_i, c_i, c_p, c_l, c_p, c_l, c_i, c_p, c_p, c_l, c_p, c_l, c_l, c_p],
    "aql_lora_gemm_fused_geglu": [c_p, c_l, c_p, c_l, c_l, c_i, c_i, c_p, c_p, c_i, c_p, c_p, c_p, c_l, c_p, c_l, c_p, c_p, c_l,
                                  c_p],
    "aql_lora_gemm_fused_grouped": [c_p, c_l, c_p, c_l, c_l, c_i, c_i, c_i, c_p, c_p, c_p, c_i, c_p, c_p, c_p, c_l, c_p, c_p, c_l,
                                    c_p],
    "aql_lora_gemm_fused_geglu_bwd": [c_p, c_l, c_p, c_l, c_l, c_i, c_i, c_p, c_p, c_i, c_p, c_p, c_l, c_p, c_l, c_p, c_p, c_p],
    "aql_gemm_bf16_geglu_bwd": [c_p, c_l, c_p, c_l, c_l, c_i, c_i, c_p, c_l, c_p, c_l, c_i, c_p, c_l, c_p, c_l, c_p, c_sz, c_p],
    "aql_lora_gemm_fused_kgroups": [c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_l, c_i, c_p, c_i, c_p, c_l, c_p, c_l, c_p, c_p, c_p],
    "aql_lora_down_grouped": [c_i, c_p, c_p, c_p, c_l, c_p, c_i, c_p, c_p, c_p],
    "aql_lora_down": [c_p, c_l, c_l, c_i, c_p, c_i, c_p, c_i, c_p, c_p, c_p, c_p, c_p],
    "aql_lora_down_splitk": [c_p, c_l, c_l, c_i, c_p, c_i, c_p, c_i, c_p, c_p, c_p, c_sz, c_p, c_sz, c_p],
    "aql_conv3x3_fwd": [c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_i, c_i, c_i, c_p, c_l, c_p, c_p, c_p, c_sz, c_p],
    "aql_conv3x3_fwd_pad": [c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_l, c_p, c_p, c_p, c_sz, c_p],
    "aql_softmax_rows": [c_p, c_l, c_l, c_i, c_f, c_p, c_l, c_p],
    "aql_quick_gelu": [c_p, c_l, c_p, c_p],
    "aql_causal_attn_small": [c_p, c_p, c_p, c_l, c_i, c_i, c_i, c_i, c_f, c_p, c_l, c_p],
    "aql_softmax_rows_bwd": [c_p, c_p, c_l, c_l, c_i, c_f, c_p, c_p],
    "aql_conv3x3_bwd_data": [c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_i, c_p, c_p, c_sz, c_p],
    "aql_gemm_bf16_grouped": [c_p, c_l, c_p, c_l, c_l, c_i, c_i, c_p, c_l, c_p, c_l, c_i, c_i, c_i, c_i, c_p, c_p, c_l, c_p, c_l, c_p, c_l,
                              c_p, c_i, c_l, c_p, c_sz, c_p],
    "aql_conv3x3_fwd_defer": [c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_i, c_i, c_i, c_p, c_l, c_p, c_p, c_p, c_sz, c_p, c_p],
    "aql_conv3x3_bwd_data_defer": [c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_i, c_p, c_p, c_sz, c_p, c_p],
    "aql_splitk_finalize": [c_p, c_i, c_l, c_i, c_p, c_p, c_l, c_i, c_p, c_l, c_p, c_l, c_p],
    "aql_groupnorm_silu_fwd_slabs": [c_p, c_i, c_p, c_p, c_l, c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_f, c_i, c_p, c_p, c_p],
    "aql_groupnorm_silu_bwd_slabs": [c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_i, c_p, c_p, c_p, c_p],
    "aql_gemm_tn_f32": [c_p, c_l, c_p, c_l, c_l, c_i, c_i, c_f, c_p, c_l, c_p],
    "aql_gemm_tn_tr_f32": [c_p, c_l, c_p, c_l, c_l, c_i, c_i, c_f, c_p, c_l, c_p],
    "aql_gemm_tn_tr_grouped": [c_p, c_i, c_i, c_i, c_i, c_p],
    "aql_groupnorm_silu_fwd": [c_p, c_i, c_i, c_i, c_p, c_p, c_f, c_i, c_p, c_p, c_p, c_p],
    "aql_groupnorm_silu_bwd": [c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_p],
    "aql_layernorm_fwd": [c_p, c_l, c_i, c_p, c_p, c_f, c_p, c_p, c_p],
    "aql_layernorm_bwd": [c_p, c_p, c_l, c_i, c_p, c_p, c_p, c_p, c_p],
    # X ldx M rps row0 S nstage | W ldw bias Adown Bup T Ts res ldr out ldo keep ln gamma beta eps stats nout ldn nout_row0 | stream
    "aql_lora_chain_fwd": [c_p, c_l, c_l, c_i, c_l, c_p, c_i] + [c_p] * 21 + [c_p],
    "aql_lora_chain_fwd_r320": [c_p, c_l, c_l, c_i, c_l, c_p, c_i] + [c_p] * 21 + [c_p],
    # dY lddy M rps S nstage | Wt ldw BupT AT dTs dT dX lddx keep | ln_x ld_lnx ln_stats ln_gamma ln_dres ld_dres ln_out ld_lnout | stream
    "aql_lora_chain_bwd": [c_p, c_l, c_l, c_i, c_p, c_i] + [c_p] * 17 + [c_p],
    "aql_geglu_fwd": [c_p, c_l, c_i, c_p, c_p],
    "aql_geglu_bwd": [c_p, c_p, c_l, c_i, c_p, c_p],
    "aql_upsample2x_bwd": [c_p, c_i, c_i, c_i, c_i, c_p, c_p],
    "aql_cat_channels": [c_p, c_p, c_l, c_i, c_i, c_p, c_p],
    "aql_split_channels": [c_p, c_l, c_i, c_i, c_p, c_p, c_p],
    "aql_add_noise": [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_p, c_p, c_p],
    "aql_mse_fwd_bwd": [c_p, c_p, c_l, c_p, c_p, c_p],
    "aql_mapper_fwd": [c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_p],
    "aql_ppft_prologue": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_l, c_p, c_p, c_p, c_p,
                          c_p, c_p, c_p],
    "aql_mapper_bwd": [c_p, c_p, c_i, c_i, c_i, c_p, c_p],
    "aql_secret_encoder_fwd": [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_p, c_p, c_p],
    "aql_cast_transpose": [c_p, c_i, c_i, c_p, c_p, c_p],
    "aql_cast_transpose_batched": [c_p, c_i, c_i, c_p],
    "aql_tn_desc_fill": [c_p, c_p, c_l, c_p, c_l, c_l, c_i, c_i, c_f, c_p, c_l, c_i],
    "aql_tntr_desc_fill": [c_p, c_p, c_l, c_p, c_l, c_l, c_i, c_i, c_f, c_p, c_l, c_i],
    "aql_tntr160_desc_fill": [c_p, c_p, c_l, c_p, c_l, c_l, c_i, c_i, c_f, c_p, c_l, c_i],
    "aql_gemm_tn_tr160_grouped": [c_p, c_i, c_i, c_i, c_i, c_p],
    "aql_gemm_tn_grouped": [c_p, c_i, c_i, c_p],
    "aql_gemm_tn_grouped_range": [c_p, c_i, c_i, c_i, c_i, c_p],
    "aql_ds_desc_fill": [c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_i],
    "aql_ds_desc_fill_ld": [c_p, c_p, c_l, c_p, c_i, c_i, c_i, c_p, c_i],
    "aql_lora_ds_grouped": [c_p, c_i, c_i, c_p],
    "aql_lora_ds": [c_p, c_p, c_i, c_i, c_i, c_p, c_p],
    "aql_wside_reduce": [c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_l, c_p, c_p],
    "aql_sumsq_f32": [c_p, c_l, c_p, c_p],
    "aql_clipnorm_adamw": [c_p, c_p, c_p, c_p, c_l, c_p, c_f, c_p, c_f, c_f, c_f, c_f, c_p, c_p],
    "aql_jpeg_mask": [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    "aql_resize_bilinear_nhwc": [c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p],
    "aql_stem_conv3x3s2_silu": [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p],
    "aql_dwconv_silu": [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p],
    "aql_avgpool_nhwc": [c_p, c_i, c_i, c_i, c_p, c_p],
    "aql_se_gate": [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_p],
    "aql_avgpool_nhwc_slabs": [c_p, c_i, c_i, c_i, c_i, c_p, c_p],
    "aql_se_gate_slabs": [c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_p],
    "aql_pwconv_f32": [c_p, c_p, c_p, c_p, c_i, c_p, c_l, c_i, c_i, c_i, c_p, c_p],
    "aql_crop_resize_bilinear": [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    "aql_gauss_blur": [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_p],
    "aql_gauss_blur2": [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_i, c_i, c_p],
    "aql_add_gauss_noise": [c_p, c_p, c_f, c_i, c_l, c_p, c_p],
    "aql_color_jiggle": [c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_p],
    "aql_rotate_bilinear": [c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_p],
    "aql_sharpness": [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p],
    "aql_transpose_bf16": [c_p, c_l, c_i, c_l, c_p, c_p],
    "aql_gemm_nt_f32_accum": [c_p, c_l, c_p, c_l, c_l, c_i, c_l, c_f, c_p, c_l, c_p, ctypes.c_size_t, c_p],
    "aql_gemm_f32": [c_p, c_l, c_l, c_p, c_l, c_l, c_p, c_p, c_l, c_l, c_i, c_l, c_p],
    "aql_bn_train_fwd": [c_p, c_p, c_p, c_l, c_i, c_f, c_f, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p],
    "aql_bn_train_bwd": [c_p, c_p, c_p, c_p, c_p, c_p, c_l, c_i, c_i, c_p, c_p, c_p, c_p, c_p],
    "aql_bn_train_fwd_res": [c_p, c_p, c_p, c_l, c_i, c_f, c_f, c_i, c_p, c_p, c_l, c_p, c_p, c_p, c_p, c_p, c_p, c_p],
    "aql_bn_train_bwd_rs": [c_p, c_p, c_p, c_p, c_p, c_p, c_l, c_i, c_i, c_p, c_l, c_p, c_p, c_p, c_p, c_p],
    "aql_dwconv_train": [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p],
    "aql_stem_train": [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p],
    "aql_chan_scale": [c_p, c_p, c_i, c_l, c_i, c_p, c_p],
    "aql_chan_reduce": [c_p, c_p, c_i, c_l, c_i, c_f, c_p, c_p],
    "aql_chan_bcast": [c_p, c_i, c_l, c_i, c_f, c_i, c_p, c_p],
    "aql_act_f32": [c_p, c_p, c_i, c_l, c_p, c_p],
    "aql_resize_bilinear_nhwc_bwd": [c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p],
    "aql_bce_logits": [c_p, c_p, c_l, c_p, c_p, c_p],
    "aql_secret_encoder_bwd": [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p],
    "aql_prvl_loss_fwd": [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p],
    "aql_prvl_loss_bwd": [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p],
    "aql_ddim_step": [c_p, c_p, c_p, c_f, c_p, c_l, c_p],
    "aql_dpmpp2m_step": [c_p, c_p, c_p, c_f, c_p, c_p, c_l, c_p],
    "aql_sampler_step": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_l, c_p],
    "aql_lpips_scale": [c_p, c_i, c_i, c_i, c_p, c_p],
    "aql_lpips_scale_bwd": [c_p, c_i, c_i, c_i, c_p, c_p],
    "aql_relu_bf16": [c_p, c_l, c_p, c_p],
    "aql_relu_bf16_bwd": [c_p, c_p, c_l, c_p, c_p],
    "aql_maxpool2x2_nhwc": [c_p, c_i, c_i, c_i, c_i, c_p, c_p],
    "aql_maxpool2x2_nhwc_bwd": [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p],
    "aql_lpips_layer": [c_p, c_p, c_p, c_i, c_l, c_i, c_p, c_p],
    "aql_lpips_layer_bwd": [c_p, c_p, c_p, c_i, c_l, c_i, c_p, c_p, c_p],
    "aql_sdpa_fwd": [c_p, c_l, c_p, c_l, c_p, c_l, c_i, c_i, c_i, c_i, c_i, c_f, c_p, c_l, c_p, c_p],
    "aql_sdpa_bwd": [c_p, c_l, c_p, c_l, c_p, c_l, c_p, c_p, c_l, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_f, c_p, c_p,
                     c_p, c_p, c_sz, c_p],
    "aql_sdpa_bwd_ex": [c_i, c_p, c_l, c_p, c_l, c_p, c_l, c_p, c_p, c_l, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_f, c_p, c_p,
                        c_p, c_l, c_l, c_p, c_sz, c_p],
    "aql_sdpa_fwd_qpre": [c_p, c_l, c_p, c_l, c_p, c_l, c_i, c_i, c_i, c_i, c_i, c_f, c_p, c_l, c_p, c_p],
    "aql_sdpa_bwd_qpre": [c_p, c_l, c_p, c_l, c_p, c_l, c_p, c_p, c_l, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_f, c_p, c_p,
                          c_p, c_p, c_sz, c_p],
    "aql_abi_version": [],
    "aql_comm_available": [],
    "aql_comm_unique_id": [c_p],
    "aql_comm_init": [c_p, c_i, c_i, c_p],
    "aql_comm_size": [c_p],
    "aql_comm_all_reduce_f32": [c_p, c_p, c_l, c_i, c_p],
    "aql_comm_reduce_scatter_f32": [c_p, c_p, c_p, c_l, c_i, c_p],
    "aql_comm_all_gather": [c_p, c_p, c_p, c_l, c_p],
    "aql_comm_broadcast": [c_p, c_p, c_l, c_i, c_p],
    "aql_comm_abort": [c_p],
    "aql_comm_destroy": [c_p],
}
ABI_VERSION = 4   # == AQL_ABI_VERSION of include/aqualora_hip.h this table was written against

_lib = None


class AqlError(RuntimeError):
    pass


def load():
    """Load the shared object once and attach argtypes.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH) and not os.environ.get("AQL_NO_AUTOBUILD"):
        # a source-only checkout: compile the HIP library in-tree (this is a build, not a fallback -- if hipcc is
        # missing or the build fails we still stop below)
        import subprocess
        res = subprocess.run(["make", "-C", os.path.join(_HERE, "csrc"), "-j8"], check=False, capture_output=True, text=True)
        if res.returncode != 0 or not os.path.exists(LIB_PATH):   # a failed build must say WHY, not surface as "not found"
            tail = "\n".join((res.stdout + res.stderr).splitlines()[-25:])
            raise AqlError(f"building {LIB_PATH} failed (make exit code {res.returncode}); there is no fallback path.  Last lines of "
                           f"the build log:\n{tail}")
    if not os.path.exists(LIB_PATH):
        raise AqlError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C aqualora_amd/csrc`). There is no fallback path.")
    lib = ctypes.CDLL(LIB_PATH)
    lib.aql_last_error.restype = ctypes.c_char_p
    lib.aql_last_error.argtypes = []
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = ABI drift, fail loudly
        fn.argtypes = argtypes
        fn.restype = c_i
    built = lib.aql_abi_version()
    if built != ABI_VERSION:
        raise AqlError(f"{LIB_PATH} was built with AQL_ABI_VERSION {built}, this binding expects {ABI_VERSION}: rebuild it "
                       "(`make -C aqualora_amd/csrc`); calling across versions would pass arguments in the wrong slots")
    _lib = lib
    return lib


def check(status, what=""):
    if status != 0:
        msg = load().aql_last_error().decode("utf-8", "replace")
        raise AqlError(f"{what} failed with status {status}: {msg}")


def ptr(t):
    """Device pointer of a tensor (or NULL for None)."""
    return None if t is None else c_p(t.data_ptr())


def stream_ptr():
    return c_p(torch.cuda.current_stream().cuda_stream)


_DEBUG_SYNC = bool(os.environ.get("AQL_DEBUG_SYNC"))   # debugging aid: synchronise after every entry point and name the one that faulted


def _debug_sync(name):
    import sys
    print(f"[aql] {name}", file=sys.stderr, flush=True)
    if not torch.cuda.is_current_stream_capturing():
        torch.cuda.synchronize()


def call(name, *args):
    lib = load()
    check(getattr(lib, name)(*args), name)
    if _DEBUG_SYNC:
        _debug_sync(name)


def call_raw(name, *args):
    """For the *_desc_fill helpers, whose return value is a workgroup count rather than a status."""
    rc = getattr(load(), name)(*args)
    if _DEBUG_SYNC and not name.endswith("_fill"):
        _debug_sync(name)
    return rc
