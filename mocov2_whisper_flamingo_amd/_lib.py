"""ctypes binding of libavsr_hip.so - the C ABI declared in include/avsr_hip.h.

The library is built in-tree (``make -C mocov2_whisper_flamingo_amd/csrc``).  There
is no fallback: if it cannot be loaded, every op raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AVSR_HIP_LIB") or os.path.join(_HERE, "libavsr_hip.so")      # override: A/B builds of the library

F32, BF16 = 0, 1
ACT_NONE, ACT_GELU, ACT_RELU, ACT_RELU_AFTER_RESIDUAL = 0, 1, 2, 3

_c = ctypes
_vp, _i64, _i32, _f32, _u64, _f64 = _c.c_void_p, _c.c_int64, _c.c_int32, _c.c_float, _c.c_uint64, _c.c_double


class ConvDesc(_c.Structure):
    _fields_ = [(n, _i32) for n in ("enabled", "H", "W", "C", "KH", "KW", "stride", "pad_h", "pad_w", "OH", "OW")]


class GemmDesc(_c.Structure):
    _fields_ = [
        ("in_dtype", _i32), ("out_dtype", _i32),
        ("M", _i64), ("N", _i64), ("K", _i64),
        ("A", _vp), ("lda", _i64), ("B", _vp), ("ldb", _i64), ("C", _vp), ("ldc", _i64),
        ("batch", _i64), ("strideA", _i64), ("strideB", _i64), ("strideC", _i64),
        ("alpha", _f32), ("scale", _vp), ("bias", _vp), ("act", _i32),
        ("residual", _vp), ("ldr", _i64), ("res_period", _i64), ("gate", _vp),
        ("col_sum", _vp), ("col_sumsq", _vp), ("conv", ConvDesc),
        ("pre", _vp), ("ldp", _i64), ("mul_gelu_grad_of", _vp), ("ld_aux", _i64), ("p_drop", _f32), ("seed", _u64),
        ("seed_delta", _vp), ("row_part", _vp),
        ("ln_in", _vp), ("ln_parts", _i64), ("ln_eps", _f32), ("ln_colsum", _vp),
        ("ln_out", _vp), ("copy_bf16", _vp), ("ld_copy", _i64),
        ("copy_lo", _vp), ("residual_hi", _vp), ("residual_lo", _vp), ("ld_res_planes", _i64),
    ]


class XattnBlock(_c.Structure):
    _fields_ = [
        ("dtype", _i32), ("B", _i64), ("Tq", _i64), ("Tk", _i64), ("d", _i64), ("H", _i64), ("d_ff", _i64),
        ("p_drop", _f32), ("seed_attn", _u64), ("seed_ff", _u64), ("seed_delta", _vp),
        ("w_in", _vp), ("w_inT", _vp), ("ld_inT", _i64), ("b_in", _vp),
        ("w_out", _vp), ("w_outT", _vp), ("ld_outT", _i64), ("b_out", _vp),
        ("w_fc1", _vp), ("w_fc1T", _vp), ("ld_fc1T", _i64), ("b_fc1", _vp),
        ("w_fc2", _vp), ("w_fc2T", _vp), ("ld_fc2T", _i64), ("b_fc2", _vp),
        ("ln1_g", _vp), ("ln1_b", _vp), ("ln2_g", _vp), ("ln2_b", _vp),
        ("gate_attn", _vp), ("gate_ff", _vp),
    ]


class XattnGrads(_c.Structure):
    _fields_ = [(n, _vp) for n in ("dw_in", "db_in", "dw_out", "db_out", "dw_fc1", "db_fc1", "dw_fc2", "db_fc2",
                                   "dln1_g", "dln1_b", "dln2_g", "dln2_b", "dgate_attn", "dgate_ff")]


# name -> (restype, argtypes); must list every symbol include/avsr_hip.h declares
SIGNATURES = {
    "avsr_version": (_c.c_char_p, []),
    "avsr_gemm_nt": (_i32, [_vp, _c.POINTER(GemmDesc)]),
    "avsr_gemm_tn": (_i32, [_vp, _vp, _i64, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _i32, _vp, _i32]),
    "avsr_layernorm_fwd": (_i32, [_vp, _i32, _i32, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _i64, _i64, _f32]),
    "avsr_layernorm_bwd": (_i32, [_vp, _i32, _i32, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _i64, _i64]),
    "avsr_layernorm_bwd_res": (_i32, [_vp, _i32, _i32, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _i64, _i64,
                                      _vp, _i64]),
    "avsr_attn_fwd": (_i32, [_vp, _i32, _i32, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp,
                             _i64, _i64, _i64, _i64, _i64, _f32, _f32, _u64, _i32]),
    "avsr_attn_fwd_workspace_bytes": (_i64, [_i64, _i64, _i64, _i64, _f32, _i32]),
    "avsr_attn_fwd_ws": (_i32, [_vp, _i32, _i32, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp,
                                _i64, _i64, _i64, _i64, _i64, _f32, _f32, _u64, _i32, _vp, _i64]),
    "avsr_attn_bwd": (_i32, [_vp, _i32, _i32, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64,
                             _vp, _i64, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _f32, _f32, _u64]),
    "avsr_gated_xattn_fwd_workspace_bytes": (_i64, [_c.POINTER(XattnBlock)]),
    "avsr_gated_xattn_bwd_workspace_bytes": (_i64, [_c.POINTER(XattnBlock)]),
    "avsr_gated_xattn_fwd": (_i32, [_vp, _c.POINTER(XattnBlock), _vp, _vp, _vp, _vp, _vp]),
    "avsr_gated_xattn_bwd": (_i32, [_vp, _c.POINTER(XattnBlock), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32,
                                    _c.POINTER(XattnGrads), _vp]),
    "avsr_logmel_dft_table_floats": (_i64, []),
    "avsr_logmel_mel_table_floats": (_i64, []),
    "avsr_logmel_tables": (_i32, [_vp, _i32, _vp, _vp]),
    "avsr_logmel_workspace_floats": (_i64, [_i64, _i64]),
    "avsr_logmel_fwd": (_i32, [_vp, _i32, _i32, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp]),
    "avsr_logmel_fwd_static": (_i32, [_vp, _i32, _i32, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _i64, _i64, _i64, _i64]),
    "avsr_video_preprocess": (_i32, [_vp, _i32, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp]),
    "avsr_video_color": (_i32, [_vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "avsr_logmel_augment_workspace_floats": (_i64, [_i64, _i64]),
    "avsr_logmel_augment_fwd": (_i32, [_vp, _i32, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp,
                                        _vp, _i64, _vp, _i64, _vp, _i64, _vp]),
    "avsr_stem3d_fwd": (_i32, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64]),
    "avsr_stem3d_fwd_indirect": (_i32, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64]),
    "avsr_store_u64": (_i32, [_vp, _vp, _u64, _vp, _vp, _i64]),
    "avsr_zero_regions": (_i32, [_vp, _i64, _vp, _vp, _vp, _vp]),
    "avsr_stem3d_frag_elems": (_i64, []),
    "avsr_stem3d_pack_frags": (_i32, [_vp, _vp, _vp]),
    "avsr_maxpool3x3s2": (_i32, [_vp, _i32, _i32, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64]),
    "avsr_avgpool_mask_fwd": (_i32, [_vp, _i32, _i32, _vp, _vp, _vp, _i64, _i64, _i64, _i64]),
    "avsr_bn_apply_avgpool_mask": (_i32, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64]),
    "avsr_bn_finalize_apply": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _f64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64,
                                      _f32, _f32, _i32]),
    "avsr_bn_finalize": (_i32, [_vp, _vp, _vp, _i64, _f64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _f32, _f32, _vp, _vp]),
    "avsr_bn_finalize_workspace_floats": (_i64, [_i64]),
    "avsr_gemm_stats_parts": (_i64, [_i32, _i64]),
    "avsr_stem3d_stats_parts": (_i64, [_i64, _i64, _i64, _i64]),
    "avsr_gram_parts": (_i64, [_i64, _i64]),
    "avsr_gram_stats": (_i32, [_vp, _vp, _i64, _i64, _i64, _vp, _vp]),
    "avsr_gram_stats_bn": (_i32, [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _i64, _vp, _vp]),
    "avsr_bn_gram_workspace_doubles": (_i64, [_i64]),
    "avsr_bn_gram_finalize": (_i32, [_vp, _vp, _vp, _i64, _i64, _f64, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp,
                                     _f32, _f32, _vp]),
    "avsr_bn_fold": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _f32]),
    "avsr_bn_apply": (_i32, [_vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32]),
    "avsr_head_loss_workspace_floats": (_i64, [_i64, _i64, _i64]),
    "avsr_head_loss_fwd": (_i32, [_vp, _vp, _i64, _i64, _i64, _i64, _vp, _i64, _i64, _vp, _vp, _f32, _i64, _i64, _vp, _vp, _vp, _i64]),
    "avsr_head_loss_bwd": (_i32, [_vp, _i32, _vp, _i64, _i64, _i64, _i64, _vp, _i64, _i64, _vp, _vp, _f32, _i64, _i64,
                                  _vp, _f32, _f32, _vp, _vp, _vp, _i64]),
    "avsr_argmax_rows": (_i32, [_vp, _vp, _i64, _i64, _i64, _vp]),
    "avsr_cast": (_i32, [_vp, _i32, _i32, _vp, _vp, _i64, _f32]),
    "avsr_transpose": (_i32, [_vp, _i32, _i32, _vp, _i64, _i64, _vp, _i64, _i64, _i64, _i64, _i64, _f32, _vp, _i64]),
    "avsr_colsum": (_i32, [_vp, _i32, _vp, _i64, _i64, _i64, _vp, _i32]),
    "avsr_gelu_fwd": (_i32, [_vp, _i32, _vp, _vp, _i64]),
    "avsr_gelu_bwd": (_i32, [_vp, _i32, _vp, _vp, _vp, _i64]),
    "avsr_gated_residual_fwd": (_i32, [_vp, _i32, _vp, _vp, _vp, _vp, _i64, _f32, _u64]),
    "avsr_gated_residual_bwd": (_i32, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _i64, _f32, _u64]),
    "avsr_sum_parts": (_i32, [_vp, _i32, _vp, _i64, _i64, _vp, _i64, _vp, _i64]),
    "avsr_cast_rows": (_i32, [_vp, _i32, _i32, _vp, _i64, _vp, _i64, _i64, _i64, _f32]),
    "avsr_length_mask": (_i32, [_vp, _vp, _i64, _i64, _vp]),
    "avsr_add3": (_i32, [_vp, _i32, _vp, _vp, _vp, _vp, _i64]),
    "avsr_add_table": (_i32, [_vp, _i32, _vp, _vp, _vp, _i64, _i64]),
    "avsr_embed_tokens": (_i32, [_vp, _i32, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _f32, _vp]),
    "avsr_sumsq_accumulate": (_i32, [_vp, _vp, _i64, _vp]),
    "avsr_clip_scale": (_i32, [_vp, _vp, _f32, _f32, _vp, _vp]),
    "avsr_adamw_step": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _f32, _i64, _vp, _i32]),
    "avsr_step_params": (_i32, [_vp, _vp, _u64, _f32, _f32, _f32, _i64]),
    "avsr_adamw_step_dev": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _f32, _f32, _f32, _f32, _vp, _i32, _vp]),
}

_lib = None


class AvsrHipError(RuntimeError):
    pass


def load():
    """Load the shared library once and attach the prototypes; raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AvsrHipError(
            f"{LIB_PATH} not found - build it with `make -C {os.path.join(_HERE, 'csrc')}`; "
            "this package has no CPU or PyTorch fallback")
    lib = _c.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


_ERR = {-1: "invalid argument", -2: "unsupported shape / alignment", -3: "HIP launch error"}


def check(rc, what):
    if rc != 0:
        raise AvsrHipError(f"{what} failed: {_ERR.get(rc, rc)}")
