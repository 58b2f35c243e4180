"""ctypes binding of libyond_hip.so (C ABI declared in include/yond_hip.h).

The product path has no CPU fallback: if the shared library is missing or a tensor is not
on a ROCm device, the callers raise.  `load()` is lazy so that importing the package (and
the host-side logic) works on a CPU-only box; the first kernel call needs the library.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("YOND_HIP_LIB", os.path.join(_HERE, "libyond_hip.so"))   # override: experiments only
_lib = None
ABI_VERSION = 8                     # include/yond_hip.h YOND_ABI_VERSION

vp, i32, f32, f64, sz, i64 = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_size_t, C.c_longlong


class YondConvDesc(C.Structure):
    _fields_ = [("src0", vp), ("src1", vp), ("C0", i32), ("C1", i32), ("N", i32), ("H", i32), ("W", i32),
                ("Ho", i32), ("Wo", i32), ("Cout", i32), ("ksize", i32), ("stride", i32), ("shuffle", i32),
                ("pre_act", i32), ("post_act", i32), ("slope", f32), ("wpk", vp), ("escale", vp),
                ("eshift", vp), ("ebatch", i32), ("res", vp), ("dst", vp), ("tn", i32), ("kc", i32), ("algo", i32),
                ("out4_w", vp), ("out4_b", vp), ("out4_x", vp), ("out4_ub", vp), ("out4_dst", vp), ("status", vp),
                ("in_fmt", i32), ("out_fmt", i32), ("res_fmt", i32), ("clk", vp), ("tile_order", i32), ("dst2", vp)]


class YondBlock0Desc(C.Structure):
    _fields_ = [("x", vp), ("in_fmt", i32), ("N", i32), ("H", i32), ("W", i32), ("w1", vp), ("w2", vp),
                ("s1", vp), ("t1", vp), ("s2", vp), ("t2", vp), ("ebatch", i32), ("dst", vp),
                ("out4_w", vp), ("out4_b", vp), ("out4_x", vp), ("out4_ub", vp), ("out4_dst", vp), ("status", vp)]


class YondFilmDesc(C.Structure):
    _fields_ = [("kind", i32), ("C", i32), ("ld", i32),
                ("w_a0", vp), ("b_a0", vp), ("w_a2", vp), ("b_a2", vp), ("w_b0", vp), ("b_b0", vp),
                ("w_b", vp), ("b_b", vp), ("cb1", vp), ("cb2", vp),
                ("s1", vp), ("t1", vp), ("s2", vp), ("t2", vp)]


# name -> argtypes, in the order of include/yond_hip.h (tests check every name is exported)
PROTOTYPES = {
    "yond_abi_version": [],
    "yond_pack_vst_norm_f32": [vp, i32, i32, vp, i32, i32, i32, i32, i32, f64, f64, f64, f64, f64, vp, vp, i32, vp, vp],
    "yond_pack_vst_norm_biaslut_f32": [vp, i32, i32, vp, i32, i32, i32, i32, f64, f64, f64, f64, f64, vp, vp, i32, vp, vp],
    "yond_bias_eval_f32": [vp, sz, vp, vp, i32, i32, i32, f64, f64, vp, vp],
    "yond_denorm_ivst_unpack_f32": [vp, i32, i32, i32, i32, i32, i32, vp, i32, f64, f64, f64, f64, f64, i32, vp],
    "yond_pack_vst_norm_batch_f32": [vp, i32, i32, i32, vp, i32, i32, i32, i32, f64, f64, f64, f64, f64, vp, vp, i32, i32, vp, vp],
    "yond_denorm_ivst_unpack_batch_f32": [vp, i32, i32, i32, i32, i32, i32, i32, vp, i32, f64, f64, f64, f64, f64, i32, vp],
    "yond_vst_elem_f32": [vp, sz, f64, f64, f64, vp, vp],
    "yond_ivst_elem_f64": [vp, sz, f64, f64, i32, vp, vp],
    "yond_bayer2rggb_f32": [vp, i32, i32, vp, vp],
    "yond_rggb2bayer_f32": [vp, i32, i32, vp, vp],
    "yond_rot90_f32": [vp, i32, i32, i32, i32, vp, vp],
    "yond_nchw4_to_nhwc4_f32": [vp, vp, i32, i32, i32, vp],
    "yond_nhwc4_to_nchw4_f32": [vp, vp, i32, i32, i32, vp],
    "yond_image_max_f32": [vp, i32, sz, vp, vp, vp],
    "yond_conv_config": [i32, i32, i32, i32, i32, i32, i32, i32, C.POINTER(i32), C.POINTER(i32)],
    "yond_pack_conv_weight_f32": [vp, i32, i32, i32, i32, i32, vp],
    "yond_conv2d_f32": [C.POINTER(YondConvDesc), vp],
    "yond_conv_wino_supported": [i32, i32],
    "yond_pack_conv_wino_weight_f32": [vp, i32, i32, i32, vp],
    "yond_pack_conv_weight_split_f32": [vp, i32, i32, i32, i32, i32, vp],
    "yond_conv_split_supported": [i32, i32, i32, i32],
    "yond_pack_conv_split_weight_f32": [vp, i32, i32, i32, i32, i32, vp],
    "yond_pack_conv_split_weight_dev_f32": [vp, i32, i32, i32, i32, i32, vp, vp, vp],
    "yond_pack_conv_split_weights_batch_dev_f32": [vp, vp, i32, vp, sz, vp, vp],
    "yond_conv_in_f32": [vp, vp, i32, i32, i32, i32, vp, vp, f32, vp, i32, vp],
    "yond_pack_conv_in_weight_f32": [vp, i32, vp],
    "yond_conv_out_f32": [vp, i32, vp, vp, vp, vp, i32, i32, i32, vp, vp],
    "yond_maxpool2_f32": [vp, i32, i32, i32, i32, vp, vp],
    "yond_film_f32": [vp, i32, vp, vp, i32, vp],
    "yond_box_stats_self1_f32": [vp, i32, i32, i32, i32, i32, vp, vp, vp, vp],
    "yond_box_stats_self2_f32": [vp, i32, i32, i32, i32, vp, vp],
    "yond_box_stats_collab_f32": [vp, vp, i32, i32, i32, i32, vp, vp, vp, vp],
    "yond_box_stats_self_stats_f32": [vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, i32, vp, vp],
    "yond_box_stats_collab_stats_f32": [vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, i32, vp, vp],
    "yond_select_ws_bytes": [i32],
    "yond_select_ranks_f32": [vp, sz, vp, i32, vp, vp, vp],
    "yond_percentiles_f32": [vp, sz, vp, i32, vp, vp, vp],
    "yond_nle_ws_bytes": [sz],
    "yond_nle_stats_f32": [vp, vp, sz, i32, vp, i32, vp, vp],
    "yond_nle_threshold_f32": [vp, sz, vp, i32, i32, vp, vp],
    "yond_nle_state_layout": [vp],
    "yond_nle_moments_f32": [vp, vp, vp, sz, vp, vp],
    "yond_nlf_occupancy_f32": [vp, vp, sz, i32, vp, i32, vp, vp],
    "yond_nlf_score3_f64": [vp, vp, vp, i32, vp, vp, vp],
    "yond_nlf_moments_f32": [vp, vp, vp, sz, vp, vp, vp],
    "yond_bias_lut_f64": [vp, i32, f64, f64, vp, vp],
    "yond_block_metrics_tiles": [i32, i32],
    "yond_block_metrics_f32": [vp, vp, i32, i32, i32, i32, vp, vp],
    "yond_clock_probe": [f64, vp, vp],
    "yond_conv_wgrad_f32": [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp],
    "yond_conv_wgrad_ws_f32": [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp, sz, vp],
    "yond_conv_wgrad_ws_bytes": [i32, i32, i32, i32, i32, i32, i32, i32, i32],
    "yond_conv_wgrad_split_ws_bytes": [i32, i32, i32, i32, i32],
    "yond_conv_wgrad_split_f32": [vp, vp, i32, i32, i32, i32, i32, vp, i32, vp, sz, vp, vp],
    "yond_colsum_f32": [vp, sz, i32, vp, vp],
    "yond_film_silu_supported": [i32],
    "yond_film_silu_f32": [vp, vp, vp, vp, i32, sz, i32, vp],
    "yond_film_silu_bwd_f32": [vp, vp, vp, vp, vp, vp, vp, i32, sz, i32, vp],
    "yond_film_mlp_fwd_f32": [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp],
    "yond_film_mlp_bwd_f32": [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp],
    "yond_film_mlp_fwd_multi_f32": [vp, i32, vp],
    "yond_gemm_split_f32": [vp, i32, sz, i32, i32, i64, i64, i32, vp, vp, i32, i32, i32, i32, vp, vp],
    "yond_film_mlp_bwd_multi_f32": [vp, i32, vp],
    "yond_silu_bwd_add_f32": [vp, vp, vp, vp, sz, vp],
    "yond_silu_f32": [vp, vp, sz, vp],
    "yond_zero_interleave_f32": [vp, i32, i32, i32, i32, i32, i32, vp, vp],
    "yond_l1_loss_f32": [vp, vp, sz, vp, vp, vp],
    "yond_charbonnier_loss_f32": [vp, vp, sz, f64, vp, vp, vp],
    "yond_adam_step_f32": [vp, vp, vp, vp, sz, f64, f64, f64, f64, i32, vp],
    "yond_adam_step_dev_f32": [vp, vp, vp, vp, sz, f64, f64, f64, vp, vp, vp],
    "yond_frame_params_f64": [vp, vp, i32, f64, f64, f64, i32, vp, vp, vp, vp],
    "yond_frame_chain_f64": [vp, vp, i32, f64, f64, f64, i32, vp, vp, vp, vp, vp, vp],
    "yond_bias_lut_dev_f64": [vp, i32, vp, vp, vp],
    "yond_bias_lut_big_scratch": [f64, f64, i32],
    "yond_bias_points_scratch": [f64, f64, i32, i32, f64, i32],
    "yond_bias_points_f64": [vp, i32, f64, f64, i32, i32, f64, vp, vp, vp, sz, i32, vp],
    "yond_bias_lut_big_f64": [vp, i32, f64, f64, vp, vp, sz, i32, vp],
    "yond_lut_ws_bytes": [i32],
    "yond_lut_table_f64": [vp, vp, i32, vp, vp, vp],
    "yond_pack_vst_norm_dev_f32": [vp, i32, i32, vp, i32, i32, i32, i32, f64, vp, vp, i32, vp, vp],
    "yond_pack_vst_norm_batch_dev_f32": [vp, i32, i32, i32, vp, i32, i32, i32, i32, f64, vp, vp, i32, vp, vp],
    "yond_pack_vst_norm_chain_f32": [vp, i32, i32, vp, i32, i32, i32, i32, f64, vp, vp, i32, vp, vp],
    "yond_denorm_ivst_unpack_dev_f32": [vp, i32, i32, i32, i32, i32, i32, vp, i32, f64, vp, i32, vp],
    "yond_denorm_ivst_unpack_batch_dev_f32": [vp, i32, i32, i32, i32, i32, i32, i32, vp, i32, f64, vp, i32, vp],
}
# experiment builds only (include/yond_hip_experiments.h): bound when the loaded library has them
EXPERIMENT_PROTOTYPES = {
    "yond_box_stats_self_fused_f32": [vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, i32, vp, vp],
    "yond_box_stats_collab_fused_f32": [vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, i32, vp, vp],
    "yond_block0_fused_f32": [C.POINTER(YondBlock0Desc), vp],          # the fused level-0 block (round 5: measured no-go)
    "yond_pack_block0_weight_f32": [vp, i32, i32, vp],
}
_SIZE_T_RET = {"yond_select_ws_bytes", "yond_nle_ws_bytes", "yond_lut_ws_bytes", "yond_bias_lut_big_scratch", "yond_bias_points_scratch",
               "yond_conv_wgrad_ws_bytes", "yond_conv_wgrad_split_ws_bytes"}


class YondHipError(RuntimeError):
    pass


def load():
    """Load libyond_hip.so (once).  Raises if it is missing -- there is no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise YondHipError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  yond_public_amd has no CPU fallback.")
    if "YOND_HIP_LIB" not in os.environ:
        # a library built from other sources than the ones beside it (an edit without a rebuild) fails here, not in a kernel
        from . import build as _build
        if os.path.exists(_build.STAMP) and _build.sources() and open(_build.STAMP).read().split()[0] != _build.source_hash():
            raise YondHipError(f"{LIB_PATH} is stale: csrc/ or include/ changed since it was built "
                               "(`python -m yond_public_amd.build`, or __graft_entry__.build())")
    lib = C.CDLL(LIB_PATH)
    for name, args in PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_size_t if name in _SIZE_T_RET else C.c_int
    for name, args in EXPERIMENT_PROTOTYPES.items():
        if hasattr(lib, name):
            fn = getattr(lib, name)
            fn.argtypes, fn.restype = args, C.c_int
    if lib.yond_abi_version() != ABI_VERSION:
        raise YondHipError(f"{LIB_PATH} has ABI version {lib.yond_abi_version()}, this package binds version {ABI_VERSION} "
                           "(YondConvDesc layout): rebuild with `python -c 'import __graft_entry__ as g; g.build()'`")
    _lib = lib
    return lib


def has(name):
    """True if the loaded library exports `name` (experiment entry points exist only in -DYOND_EXPERIMENTS builds)."""
    return hasattr(load(), name)


def check(rc, what):
    if rc != 0:
        kind = {-1: "YOND_EINVAL", -2: "YOND_EUNSUPPORTED"}.get(rc, f"hipError {rc}")
        raise YondHipError(f"{what} failed: {kind}")


def require_cuda(t, name="tensor"):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise YondHipError(f"{name} must be a float32 tensor on a ROCm device (the HIP path has no CPU fallback)")
    if t.dtype != torch.float32:
        raise YondHipError(f"{name} must be float32, got {t.dtype}")
    if not t.is_contiguous():
        raise YondHipError(f"{name} must be contiguous")
    return t


class GemmSrc(C.Structure):
    """YondGemmSrc of include/yond_hip.h (one source of yond_gemm_split_f32)."""
    _fields_ = [('x', C.c_void_p), ('w', C.c_void_p), ('sk_lo', C.c_longlong), ('sk_hi', C.c_longlong),
                ('ld', C.c_int), ('k', C.c_int), ('kblk', C.c_int), ('k_real', C.c_int)]


class FilmMlpDesc(C.Structure):
    """YondFilmMlpDesc of include/yond_hip.h (one guided block's sigma-MLPs for the multi-block entries)."""
    _fields_ = [(k, C.c_void_p) for k in ('t', 'w1', 'b1', 'W2', 'b2', 'W3', 'b3', 'tk', 'tb', 'dtk', 'dtb', 'scratch',
                                          'dw1', 'db1', 'dW2', 'db2', 'dW3', 'db3')] + [(k, C.c_int) for k in ('B', 'C', 'ld', 'pad_')]


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
