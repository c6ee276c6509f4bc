"""ctypes binding of libstyler_hip.so (the C ABI declared in include/styler_hip.h).

The product path has NO CPU fallback: importing this module without the built library raises, and
every op raises on a non-zero return code."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("STYLER_LIB") or os.path.join(_HERE, "libstyler_hip.so")     # STYLER_LIB: A/B builds

P = ctypes.c_void_p
I = ctypes.c_int
I64 = ctypes.c_int64
F = ctypes.c_float

_SIGS = {
    "styler_abi_version": [],
    "styler_conv_gemm": [P, I64, P, P, P, P, I64, P, I64, I, I, I, I, I, I, I, P, P, I64, I, P],
    "styler_conv_gemm_pad": [P, I64, P, P, P, P, I64, P, I64, I, I, I, I, I, I, I, I, P],
    "styler_leaky_sum": [P, P, P, P, I64, F, F, P],
    "styler_conv_gemm_variant": [I, I, I, I, I, I],
    "styler_conv_gemm_engine": [I, I, I, I, I, I, I, I64, I],
    "styler_gemm_set_trace": [P],
    "styler_wave_sum_selftest": [P, P, P, I, P],
    "styler_cast_bf16": [P, P, I64, P],
    "styler_cast_from_bf16": [P, P, I64, P],
    "styler_repack_conv_weight": [P, P, I, I, I, I, I, P],
    "styler_attention_fwd": [P, P, P, I, I, P, P, P],
    "styler_attention_fwd_bf16": [P, P, P, I, I, P, P, P],
    "styler_attention_fwd_x3": [P, P, P, I, I, P, P, P],
    "styler_attention_bwd_x3": [P, P, P, P, P, P, I, I, P, P, P],
    "styler_attention_fwd_bf16_io": [P, P, P, I, I, P, P, I, P],
    "styler_attention_bwd_bf16": [P, P, P, P, P, P, I, I, P, P, I, P],
    "styler_add_layernorm": [P, I64, P, I64, P, P, P, I64, P, P, P, I, I, I, P, F, ctypes.c_uint64, F, ctypes.c_uint64, P, I64, P, I64, I, P],
    "styler_linear_ln": [P, I64, I, P, P, P, I64, P, P, P, I64, P, I64, P, I64, I, I, P, F, ctypes.c_uint64, I, P],
    "styler_linear_ln_ok": [I64, I, I, I64],
    "styler_linear_ln_set_trace": [P],
    "styler_groupnorm_relu": [P, I64, P, P, P, I64, P, P, I, I, I, I, I, P],
    "styler_bn_fold": [P, P, P, P, P, P, P, I, P],
    "styler_batchnorm_train": [P, P, P, P, P, P, P, P, P, I, I64, I, I, F, ctypes.c_uint64, I, I, P],
    "styler_embed_pos": [P, P, P, P, I, I, I, P],
    "styler_add_pos": [P, I64, P, P, I, I, I, P],
    "styler_gather_rows": [P, P, P, I64, I, P],
    "styler_sinusoid_table": [P, I, I, P],
    "styler_onehot_conv5": [P, P, P, P, I64, P, P, I, I, I, P],
    "styler_mel_calibrate": [P, I64, P, I64, P, P, I, I, I, I, P],
    "styler_mel_calibrate_io": [P, I64, P, I64, P, P, I, I, I, I, I, P],
    "styler_lstm_bidir": [P, P, P, P, P, I, I, I, P],
    "styler_lstm_bidir_multi": [P, I, I, I, P],
    "styler_set_dropout_counter": [P],
    "styler_strided_copy_multi": [P, I, I64, P],
    "styler_step_begin": [P, I64, P, I64, P, P],
    "styler_strided_copy_multi_map": [P, I, I64, P, P],
    "styler_wgrad_group_desc": [P, P, I64, P, I64, P, P, I, I, I, I, I, I, I, P, P, P, I, I],
    "styler_wgrad_group": [P, I, I, I, P],
    "styler_pack_plan": [P, I, I, P, P, P, P, P],
    "styler_pack_rows": [P, I64, P, I64, P, P, I, I, I, I, P],
    "styler_unpack_rows": [P, I64, P, I64, P, I, I, I, I, P],
    "styler_conv_gemm_packed": [P, I64, P, P, P, P, I64, P, I64, I, I, I, I, I, I, P, P, P, I64, I, P],
    "styler_wgrad_packed": [P, I64, P, I64, P, P, I64, I64, I64, I, I, I, I, I, P, I, P, P, P, I, P],
    "styler_lstm_bidir_bwd_multi": [P, I, I, I, P],
    "styler_lstm_bidir_multi_mfma": [P, I, I, I, I, P],
    "styler_lstm_bidir_bwd_multi_mfma": [P, I, I, I, I, P],
    "styler_aug_classifier_tail": [P, P, P, P, P, P, I, I, P],
    "styler_duration_scan": [P, I, P, F, P, P, P, I, I, P],
    "styler_length_regulate": [P, I64, P, P, I64, P, I, I, I, I, P],
    "styler_bucket_embed_add": [P, I64, P, I64, P, F, P, F, P, P, P, P, P, P, I64, P, P, P, I, I, P],
    "styler_add2": [P, I64, P, I64, P, I64, I64, I, P],
    "styler_copy_rows_multi": [P, I, P],
    "styler_masked_err_mean_multi": [P, I, P],
    "styler_masked_err_bwd_multi": [P, I, P],
    "styler_split3_bf16": [P, I64, P, I64, I, P, I, P],
    "styler_split3_multi": [P, I, P],
    "styler_lo_part": [P, I64, P, I64, I, P, P],
    "styler_add_rowvec": [P, I64, P, I64, P, I64, I, I, I, P],
    "styler_length_mask": [P, P, I, I, P],
    "styler_style_cat": [P, P, P, P, P, P, P, P, P, I, I, P],
    "styler_add3": [P, I64, P, I64, P, I64, P, I64, I64, I, P],
    "styler_length_mask2": [P, P, I, I, P, P, I, I, P],
    "styler_masked_err_sum": [P, I64, P, I64, P, I, I, I, I, P, P],
    "styler_act_bwd": [P, I64, P, I64, P, I64, I, I, I, I, P, P],
    "styler_wgrad": [P, I64, P, I64, P, P, P, I64, I64, I64, I, I, I, I, I, I, I, P, I, I, P],
    "styler_wgrad_splits": [I, I, I, I, I, I, I],
    "styler_wgrad_dma_config": [I, I],
    "styler_wgrad_tune": [I, I],
    "styler_wgrad_x3cat_ok": [I, I, I, I],
    "styler_wgrad_splits_io": [I, I, I, I, I, I, I, I],
    "styler_wgrad_workspace_bytes_io": [I, I, I, I, I, I, I, I],
    "styler_wgrad_reduce_multi": [P, I, I64, P],
    "styler_wgrad_reduce_multi_map": [P, I, I64, P, P],
    "styler_wgrad_reduce_blocks": [I, I, I, I64, I64],
    "styler_wgrad_workspace_bytes": [I, I, I, I, I, I, I],
    "styler_colsum": [P, I64, P, P, I64, I, P],
    "styler_repack_weight_bwd": [P, P, I, I, I, I, P],
    "styler_attention_bwd": [P, P, P, P, P, P, I, I, P, P, P],
    "styler_layernorm_bwd": [P, I64, P, I64, P, P, P, I64, P, P, P, P, P, P, I, I, I, P, F, ctypes.c_uint64, F, ctypes.c_uint64, P, I64, I, I, P],
    "styler_conv_gemm_group": [P, I, P],
    "styler_act_bwd_multi": [P, I, P],
    "styler_gemm256_config": [I, I],
    "styler_gemm256_policy": [I, I],
    "styler_gemm256_height": [I, I],
    "styler_conv_gemm_engine2": [I, I, I, I, I, I, I, I64, I, I, I],
    "styler_gemm_n96_config": [I, I],
    "styler_gemm_small_split_config": [I],
    "styler_groupnorm_fused_rows": [I],
    "styler_conv_gemm_workspace_bytes": [I, I, I, I, I, I, I, I, I64, I, I],
    "styler_gemm_set_workspace": [P, I64],
    "styler_gemm_set_counters": [P, I64],
    "styler_gemm256_fixup": [I],
    "styler_set_x3_out": [P, I],
    "styler_bn_workspace_doubles": [I64, I, I],
    "styler_fold_replicas": [P, P, P, P, P, P, I, I, P],
    "styler_groupnorm_relu_bwd": [P, I64, P, I64, P, P, P, P, I64, P, P, P, I, I, I, I, I, P],
    "styler_batchnorm_bwd": [P, P, P, P, P, P, P, P, P, P, I, I64, I, I, P, F, ctypes.c_uint64, I, I, P],
    "styler_embed_bwd": [P, P, I64, P, I, I, I, P],
    "styler_embed_bwd_det": [P, P, I64, P, I, I, I, I, P],
    "styler_onehot_expand": [P, P, I64, P],
    "styler_mel_calibrate_bwd": [P, I64, P, I64, P, P, I, I, I, I, P],
    "styler_mel_calibrate_bwd_io": [P, I64, P, I64, P, P, I, I, I, I, I, P],
    "styler_lstm_bidir_bwd": [P, P, P, P, P, I, I, I, P],
    "styler_aug_classifier_tail_bwd": [P, P, P, P, P, P, P, P, P, P, P, I, I, P],
    "styler_aug_classifier_tail_bwd_io": [P, P, P, P, P, P, P, P, P, P, P, I, I, I, P],
    "styler_aug_classifier_tail_slots": [I, I],
    "styler_length_regulate_bwd": [P, I64, P, P, I64, I, I, I, I, P],
    "styler_bucket_embed_bwd": [P, P, P, P, P, I, I, P],
    "styler_bucket_embed_bwd_slots": [P, P, P, P, P, I, I, P],
    "styler_bucket_embed_slices": [],
    "styler_rowsum": [P, I64, P, I64, I, I, I, I, P],
    "styler_masked_err_bwd": [P, I64, P, I64, P, P, P, I, I, I, I, P, P],
    "styler_nll": [P, P, P, P, P, I, P],
    "styler_masked_err_mean": [P, I64, P, I64, P, P, I, I, I, I, P, P],
    "styler_nll3": [P, P, P, P, I, P, P, P, I, P],
    "styler_weighted_sum": [P, P, I, P, P],
    "styler_scale_weights": [P, P, I, P, P],
    "styler_loss_tail": [P, P, I, P, P, I, P, I, I, P, P],
    "styler_loss_tail_bwd": [P, P, I, P, P, I, P, I, I, P, P, P],
    "styler_dropout": [P, I64, P, I64, I64, I, F, ctypes.c_uint64, P],
    "styler_sumsq": [P, I64, P, P],
    "styler_adam_step": [P, P, P, P, I64, P, F, F, F, F, F, I, F, P],
    "styler_stft_mel_workspace_bytes": [I, I],
    "styler_stft_mel": [P, I64, P, P, P, P, P, P, P, I, I, I, P],
    "styler_ds_vad_bounds": [P, I64, P, I, I, P, P, P],
    "styler_ds_fbank_workspace_bytes": [I],
    "styler_ds_fbank": [P, I64, P, P, I, P, P, P, P, I, I, P],
    "styler_ds_conv1": [P, P, P, P, P, I, I, I, P],
    "styler_ds_rows": [P, P, I, I, I, I, I, I, I, P],
    "styler_ds_crelu_add": [P, P, P, I64, P],
    "styler_l2_normalize_rows": [P, P, I, I, P],
    "styler_stft_mel_varlen": [P, I64, P, P, P, P, P, P, P, F, F, P, P, P, I, I, I, P],
}


class LstmDesc(ctypes.Structure):
    _fields_ = [("gx", P), ("w_hh", P), ("out", P), ("cell_out", P), ("gates_out", P), ("H", ctypes.c_int32),
                ("_pad", ctypes.c_int32)]


class WgradGroupDesc(ctypes.Structure):
    _fields_ = [("dz", ctypes.c_uint64), ("x", ctypes.c_uint64), ("db", ctypes.c_uint64), ("db2", ctypes.c_uint64),
                ("ws", ctypes.c_uint64), ("counts", ctypes.c_uint64), ("chunktab", ctypes.c_uint64),
                ("lddz", ctypes.c_int64), ("ldx", ctypes.c_int64)] + \
               [(k, ctypes.c_int32) for k in ("B", "L", "n", "cin", "pad_left", "ct", "cpi", "cps", "tiles", "splits",
                                              "block_start", "nblocks", "variant", "kw")]


class GemmProblem(ctypes.Structure):
    _fields_ = [(k, ctypes.c_void_p) for k in ("x", "w", "scale", "shift", "res", "y", "len")] + \
               [(k, ctypes.c_int64) for k in ("ldx", "ldres", "ldy")] + \
               [(k, ctypes.c_int32) for k in ("B", "L", "cin", "n", "act", "flags")]


class ActSeg(ctypes.Structure):
    _fields_ = [(k, ctypes.c_void_p) for k in ("dy", "y", "dz")] + [(k, ctypes.c_int64) for k in ("lddy", "ldy", "rows")] + \
               [(k, ctypes.c_int32) for k in ("C", "act")]


class MaskedTerm(ctypes.Structure):
    _fields_ = [(k, ctypes.c_void_p) for k in ("a", "b", "acc", "mean", "len", "gscale", "da")] + \
               [("lda", ctypes.c_int64), ("ldb", ctypes.c_int64)] + [(k, ctypes.c_int32) for k in ("B", "L", "C", "kind")]


class CopySeg(ctypes.Structure):
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("ld_src", ctypes.c_int64), ("ld_dst", ctypes.c_int64),
                ("rows", ctypes.c_int64), ("C", ctypes.c_int32), ("_pad", ctypes.c_int32)]


class CopyDesc(ctypes.Structure):
    _fields_ = [("src", ctypes.c_uint64), ("src2", ctypes.c_uint64), ("dst", ctypes.c_uint64)] + \
               [(k, ctypes.c_int64) for k in ("ss0", "ss1", "ss2", "ds0", "ds1", "ds2", "block_start")] + \
               [(k, ctypes.c_int32) for k in ("d0", "d1", "d2", "flags")]


class LstmBwdDesc(ctypes.Structure):
    _fields_ = [("dout", P), ("gates", P), ("cell", P), ("w_hh", P), ("dgp", P), ("H", ctypes.c_int32),
                ("_pad", ctypes.c_int32)]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C styler_amd/csrc`).  styler_amd has no CPU fallback.")
    # torch first: its wheel carries its own HIP runtime (libamdhip64), and device pointers only mean something to the runtime
    # that made them.  Loaded before torch, this library pulled in /opt/rocm's copy and every launch on a torch tensor failed with
    # hipErrorNoDevice (seen with `python __graft_entry__.py smoke`: build() imports the package before smoke() imports torch).
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in _SIGS.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.argtypes = argtypes
        fn.restype = ctypes.c_int64 if name.endswith(("_bytes", "_blocks", "_bytes_io", "_doubles")) else ctypes.c_int
    return lib


lib = _load()
ABI_VERSION = lib.styler_abi_version()
EXPORTED = tuple(_SIGS)
