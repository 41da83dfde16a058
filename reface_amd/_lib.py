"""ctypes binding of the C-ABI in include/reface_hip.h (libreface_hip.so, built by reface_amd/build.py).

There is NO fallback: if the shared library is missing or fails to load, importing any compute
path raises (the product never routes through PyTorch eager ops or the CPU oracle).
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# REFACE_HIP_LIB selects another build of the same library (A/B kernel experiments); there is still no non-HIP fallback.
LIB_PATH = os.environ.get("REFACE_HIP_LIB") or os.path.join(HERE, "lib", "libreface_hip.so")

RF_F32, RF_BF16, RF_FP8_E4M3, RF_BF16X3, RF_F16 = 0, 1, 2, 3, 4
ACT_NONE, ACT_GEGLU, ACT_SILU, ACT_QUICK_GELU, ACT_GELU, ACT_RELU, ACT_SIGMOID, ACT_PRELU = 0, 1, 2, 3, 4, 5, 6, 7


class ConvGemmDesc(C.Structure):
    """Mirror of ``rf_conv_gemm_desc`` (include/reface_hip.h)."""
    _fields_ = [
        ("dtype", C.c_int32), ("out_dtype", C.c_int32),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("src0", C.c_void_p), ("src1", C.c_void_p),
        ("C0", C.c_int32), ("C1", C.c_int32), ("ld0", C.c_int32), ("ld1", C.c_int32),
        ("Hin", C.c_int32), ("Win", C.c_int32), ("Hout", C.c_int32), ("Wout", C.c_int32),
        ("KH", C.c_int32), ("KW", C.c_int32), ("stride", C.c_int32), ("pad_t", C.c_int32), ("pad_l", C.c_int32),
        ("ups", C.c_int32),
        ("W", C.c_void_p), ("ldw", C.c_int32), ("bias", C.c_void_p), ("rowvec", C.c_void_p),
        ("rows_per_sample", C.c_int32), ("ldv", C.c_int32),
        ("residual", C.c_void_p), ("ldr", C.c_int32), ("act", C.c_int32),
        ("out", C.c_void_p), ("ldo", C.c_int32), ("alpha", C.c_float),
        ("batch", C.c_int32),
        ("sA", C.c_int64), ("sW", C.c_int64), ("sO", C.c_int64), ("sR", C.c_int64),
        ("act_vec", C.c_void_p), ("korder", C.c_int32),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64),
        ("gn_rows", C.c_int32),
        ("gn_part0", C.c_void_p), ("gn_cpg0", C.c_int32), ("gn_coff0", C.c_int32), ("gn_slot0", C.c_int32), ("gn_nchunks0", C.c_int32),
        ("gn_part1", C.c_void_p), ("gn_cpg1", C.c_int32), ("gn_coff1", C.c_int32), ("gn_slot1", C.c_int32), ("gn_nchunks1", C.c_int32),
        ("w_dtype", C.c_int32), ("wscale", C.c_void_p),
        ("ascale", C.c_void_p), ("as_ld", C.c_int32),
        ("oscale", C.c_void_p), ("os_ld", C.c_int32),
        ("ln_stats_out", C.c_void_p), ("ln_out_parts", C.c_int32),
        ("ln_stats_in", C.c_void_p), ("ln_in_parts", C.c_int32), ("ln_in_cols", C.c_int32), ("ln_eps", C.c_float), ("ln_u", C.c_void_p),
        ("w_sample_stride", C.c_int64),
    ]


class FfnDesc(C.Structure):
    """Mirror of ``rf_ffn_desc`` (include/reface_hip.h)."""
    _fields_ = [
        ("x", C.c_void_p), ("ldx", C.c_int32),
        ("w1p", C.c_void_p), ("b1p", C.c_void_p),
        ("w2q", C.c_void_p), ("b2", C.c_void_p),
        ("residual", C.c_void_p), ("ldr", C.c_int32),
        ("out", C.c_void_p), ("ldo", C.c_int32),
        ("M", C.c_int32), ("C", C.c_int32),
        ("ln_eps", C.c_float),
        ("wpo", C.c_void_p), ("bpo", C.c_void_p),
        ("res2", C.c_void_p), ("ldr2", C.c_int32), ("res2_rows", C.c_int32),
        ("gn_rows", C.c_int32),
        ("gn_part0", C.c_void_p), ("gn_cpg0", C.c_int32), ("gn_coff0", C.c_int32), ("gn_slot0", C.c_int32), ("gn_nchunks0", C.c_int32),
        ("gn_part1", C.c_void_p), ("gn_cpg1", C.c_int32), ("gn_coff1", C.c_int32), ("gn_slot1", C.c_int32), ("gn_nchunks1", C.c_int32),
        ("dtype", C.c_int32),
        ("wo", C.c_void_p), ("bo", C.c_void_p),
        ("ctx", C.c_void_p), ("ldc", C.c_int32), ("rows_per_sample0", C.c_int32),
        ("res0", C.c_void_p), ("ldr0", C.c_int32), ("front_rows", C.c_int32),
        ("x1", C.c_void_p), ("ldx1", C.c_int32),
    ]


class AttnInDesc(C.Structure):
    """Mirror of ``rf_attn_in_desc`` (include/reface_hip.h)."""
    _fields_ = [
        ("x", C.c_void_p), ("ldx", C.c_int32),
        ("wpi", C.c_void_p), ("w_sample_stride", C.c_int64),
        ("rowvec", C.c_void_p), ("ldv", C.c_int32), ("rows_per_sample", C.c_int32),
        ("tok", C.c_void_p), ("ldt", C.c_int32),
        ("wqkv", C.c_void_p), ("bqkv", C.c_void_p),
        ("qkv", C.c_void_p), ("ldq", C.c_int32),
        ("M", C.c_int32), ("C", C.c_int32),
        ("ln_eps", C.c_float),
        ("dtype", C.c_int32),
    ]


class StemDesc(C.Structure):
    """Mirror of ``rf_stem_desc`` (include/reface_hip.h)."""
    _fields_ = [
        ("x", C.c_void_p), ("ldx", C.c_int32),
        ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32),
        ("w", C.c_void_p), ("bias", C.c_void_p),
        ("out", C.c_void_p), ("ldo", C.c_int32),
        ("dup_off", C.c_longlong),
        ("gn_part0", C.c_void_p), ("gn_cpg0", C.c_int32), ("gn_coff0", C.c_int32), ("gn_slot0", C.c_int32), ("gn_nchunks0", C.c_int32),
        ("gn_part1", C.c_void_p), ("gn_cpg1", C.c_int32), ("gn_coff1", C.c_int32), ("gn_slot1", C.c_int32), ("gn_nchunks1", C.c_int32),
        ("gn_part2", C.c_void_p), ("gn_cpg2", C.c_int32), ("gn_coff2", C.c_int32), ("gn_slot2", C.c_int32), ("gn_nchunks2", C.c_int32),
        ("dtype", C.c_int32),
    ]


_SIGS = {
    "rf_last_error": (C.c_char_p, []),
    "rf_version": (C.c_int, []),
    "rf_conv_gemm": (C.c_int, [C.POINTER(ConvGemmDesc), C.c_void_p]),
    "rf_conv_gemm_plan": (C.c_int, [C.POINTER(ConvGemmDesc), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "rf_conv_gemm_plan2": (C.c_int, [C.POINTER(ConvGemmDesc), C.POINTER(C.c_int32)]),
    "rf_ffn_geglu": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                               C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "rf_ffn_block": (C.c_int, [C.POINTER(FfnDesc), C.c_void_p]),
    "rf_attn_in": (C.c_int, [C.POINTER(AttnInDesc), C.c_void_p]),
    "rf_quantize_fp8_rows": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rf_groupnorm_stats": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "rf_groupnorm_finalize": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "rf_groupnorm_apply": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "rf_groupnorm_fold_linear": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rf_gn_silu_conv3x3_small": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int,
                                           C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_longlong, C.c_void_p]),
    "rf_conv3x3_stem": (C.c_int, [C.POINTER(StemDesc), C.c_void_p]),
    "rf_quantize_fp8_act": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "rf_groupnorm_apply_fp8": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float,
                                         C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "rf_layernorm_fp8": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                   C.c_void_p]),
    "rf_layernorm": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_int,
                               C.c_void_p, C.c_int, C.c_void_p]),
    "rf_attention": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                               C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_float, C.c_void_p]),
    "rf_softmax_rows": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "rf_ddim_pack_input": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "rf_ddim_update": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                 C.c_void_p, C.c_void_p]),
    "rf_nchw_to_nhwc": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "rf_nhwc_to_nchw": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "rf_cast": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]),
    "rf_split_bf16": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "rf_timestep_embedding": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rf_silu_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "rf_to_image": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "rf_u8_to_norm": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rf_label_mask": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "rf_mul_mask": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "rf_compose_outputs_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                        C.c_int64, C.c_void_p]),
    "rf_resize_u8_linear": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "rf_channel_affine": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                    C.c_int64, C.c_int, C.c_void_p]),
    "rf_spatial_mean": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "rf_se_scale_add": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                  C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "rf_adaptive_avgpool": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                      C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "rf_bilinear_resize": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                     C.c_void_p, C.c_void_p]),
    "rf_clip_tokens": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "rf_l2norm_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "rf_combine3": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_int64, C.c_void_p]),
    "rf_gaussian_sample": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
}

EXPORTS = tuple(_SIGS)
_lib = None


class RefaceHipError(RuntimeError):
    pass


def load():
    """Load libreface_hip.so (once) and attach the prototypes.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm bundles its own HIP runtime; it must be loaded first so that libreface_hip.so binds to the
    # SAME libamdhip64 instance (device pointers and streams are shared with torch).  Loading this library
    # before torch pulls in /opt/rocm's runtime instead and every launch fails with "no ROCm-capable device".
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise RefaceHipError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -m reface_amd.build` "
            "(or __graft_entry__.build()). There is no CPU / eager fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)       # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().rf_last_error()
        raise RefaceHipError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")
