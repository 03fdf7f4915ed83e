"""ctypes binding of libuegan_hip.so (the C ABI declared in include/uegan_hip.h).

The library is built in-tree by `uegan_amd/csrc/build.sh` (hipcc --offload-arch=gfx950).  There is no CPU
fallback: if the shared object is missing the first op raises.  (`_inject_for_tests` exists so the test-suite can
run the SAME kernel sources compiled against the fiber-based HIP emulator in tests/emu; the product never calls it.)
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libuegan_hip.so")
# the same sources built with -DUEGAN_HALF_FP16: the 16-bit storage format is IEEE fp16 instead of bfloat16 (csrc/common.h); same ABI,
# dtype code 1 then means fp16.  Selected by uegan_amd.set_compute_dtype(torch.float16) through use_half_format().
LIB_PATH_F16 = os.path.join(_HERE, "libuegan_hip_f16.so")

_lib = None
_lib_f16 = None
_use_f16 = False
_emulated = False

c_int, c_i64, c_f32, c_vp, c_sz = C.c_int, C.c_int64, C.c_float, C.c_void_p, C.c_size_t


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("dtype", "B", "H", "W", "C1", "C2", "Ho", "Wo", "Cout", "KH", "KW", "stride", "pad",
                                          "pad_mode", "act", "Cin_w", "Cout_w", "Cin_total", "scale_group")]


class ProfileEntry(C.Structure):
    _fields_ = [("name", C.c_char * 96), ("launches", c_i64), ("total_ms", C.c_double), ("total_flops", C.c_double),
                ("total_bytes", C.c_double)]


class SnLayer(C.Structure):
    _fields_ = [("w", c_vp), ("u", c_vp), ("v", c_vp), ("sigma", c_vp), ("inv_sigma", c_vp), ("u_hist", c_vp), ("v_hist", c_vp), ("tmp", c_vp),
                ("rows", C.c_int32), ("cols", C.c_int32)]


class PackEntry(C.Structure):
    _fields_ = [("w_oihw", c_vp), ("w_ohwi", c_vp), ("w_ihwo", c_vp), ("start", c_i64), ("Cout", c_int), ("Cin", c_int), ("Cin_total", c_int),
                ("KH", c_int), ("KW", c_int), ("Cout_pad", c_int), ("Cin_pad", c_int), ("Kp", c_int), ("Kp2", c_int), ("flags", c_int),
                ("w_ohwi_lo", c_vp)]


class ConvEx(C.Structure):
    """uegan_conv_ex: the extras of uegan_conv2d_fwd_ex (hi + lo pairs, product / residual epilogues, moments)"""
    _fields_ = [("x1_lo", c_vp), ("x2_lo", c_vp), ("w_lo", c_vp), ("y_lo", c_vp), ("mul", c_vp), ("mul_lo", c_vp), ("y_mul", c_vp), ("y_mul_lo", c_vp),
                ("res_x", c_vp), ("res_x2", c_vp), ("res_out", c_vp), ("res_out2", c_vp), ("mean", c_vp), ("rstd", c_vp), ("stats_workspace", c_vp),
                ("stats_workspace_bytes", c_sz), ("eps", c_f32), ("res_split", C.c_int32), ("w_interleaved", C.c_int32), ("reserved", C.c_int32)]


class AdamTensor(C.Structure):
    _fields_ = [("p", c_vp), ("g", c_vp), ("m", c_vp), ("v", c_vp), ("n", c_i64)]


# name -> (restype, argtypes).  Every symbol include/uegan_hip.h declares must be listed here
# (tests/test_abi.py cross-checks the header against this table and against the built library).
SIGNATURES = {
    "uegan_version": (c_int, []),
    "uegan_last_error": (C.c_char_p, []),
    "uegan_set_conv_impl": (c_int, [c_int]),
    "uegan_set_tuning": (c_int, [c_int, c_int, c_vp]),
    "uegan_selftest_mfma": (c_int, [c_vp, c_vp]),
    "uegan_profile_begin": (c_int, [c_int]),
    "uegan_profile_end": (c_int, [C.POINTER(ProfileEntry), c_int, C.POINTER(c_int)]),
    "uegan_packed_k": (c_i64, [c_i64]),
    "uegan_pack_weights": (c_int, [c_int, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp]),
    "uegan_pack_weights_slice": (c_int, [c_int, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp]),
    "uegan_pack_weights_pair": (c_int, [c_int, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_int, c_vp]),
    "uegan_pack_weights_multi": (c_int, [c_int, c_vp, c_int, c_i64, c_vp]),
    "uegan_conv2d_fwd_ex_workspace_bytes": (c_sz, [C.POINTER(ConvDesc), C.POINTER(ConvEx)]),
    "uegan_conv2d_fwd_ex": (c_int, [C.POINTER(ConvDesc), C.POINTER(ConvEx), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, C.POINTER(c_int), c_vp]),
    "uegan_nchw_to_nhwc_pair": (c_int, [c_int, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, C.POINTER(c_f32), C.POINTER(c_f32), c_vp]),
    "uegan_instnorm_apply_pair": (c_int, [c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp]),
    "uegan_conv2d_fwd": (c_int, [C.POINTER(ConvDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "uegan_conv2d_fwd_splitk_workspace_bytes": (c_sz, [C.POINTER(ConvDesc)]),
    "uegan_conv2d_fwd_splitk": (c_int, [C.POINTER(ConvDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "uegan_conv2d_fwd_pool": (c_int, [C.POINTER(ConvDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "uegan_conv2d_fwd_pool_part": (c_int, [C.POINTER(ConvDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_vp]),
    "uegan_conv2d_fwd_pool_idx": (c_int, [C.POINTER(ConvDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_vp]),
    "uegan_maxpool2x2_fwd_idx": (c_int, [c_int, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "uegan_maxpool2x2_bwd_idx": (c_int, [c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "uegan_conv2d_dgrad": (c_int, [C.POINTER(ConvDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "uegan_conv2d_dgrad_workspace_bytes": (c_sz, [C.POINTER(ConvDesc)]),
    "uegan_conv2d_dgrad_ws": (c_int, [C.POINTER(ConvDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "uegan_conv2d_dgrad_act": (c_int, [C.POINTER(ConvDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_int, c_vp, c_vp]),
    "uegan_conv2d_fwd_stats_workspace_bytes": (c_sz, [C.POINTER(ConvDesc)]),
    "uegan_conv2d_fwd_stats": (c_int, [C.POINTER(ConvDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_f32, c_vp, c_sz, C.POINTER(c_int), c_vp]),
    "uegan_instnorm_apply": (c_int, [c_int, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp]),
    "uegan_conv2d_dgrad_padded": (c_int, [C.POINTER(ConvDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, C.POINTER(c_int), c_vp]),
    "uegan_conv2d_dgrad_padded_bytes": (c_sz, [C.POINTER(ConvDesc)]),
    "uegan_conv2d_wgrad_workspace_bytes": (c_sz, [C.POINTER(ConvDesc)]),
    "uegan_conv2d_wgrad": (c_int, [C.POINTER(ConvDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "uegan_conv2d_wgrad_acc": (c_int, [C.POINTER(ConvDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_int, c_vp]),
    "uegan_act_bwd": (c_int, [c_int, c_int, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "uegan_act_bwd2": (c_int, [c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "uegan_act_bwd3": (c_int, [c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "uegan_nchw_to_nhwc": (c_int, [c_int, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, C.POINTER(c_f32), C.POINTER(c_f32), c_vp]),
    "uegan_nhwc_to_nchw": (c_int, [c_int, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, C.POINTER(c_f32), c_vp]),
    "uegan_residual_clamp_fwd": (c_int, [c_int, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "uegan_residual_clamp_bwd": (c_int, [c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "uegan_residual_clamp_bwd_act": (c_int, [c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "uegan_mul_bwd_act": (c_int, [c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "uegan_mul_fwd": (c_int, [c_int, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "uegan_mul_bwd": (c_int, [c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "uegan_upsample2x_fwd": (c_int, [c_int, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "uegan_upsample2x_bwd": (c_int, [c_int, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "uegan_maxpool2x2_fwd": (c_int, [c_int, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "uegan_maxpool2x2_bwd": (c_int, [c_int, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "uegan_maxpool2x2_bwd_act": (c_int, [c_int, c_int, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "uegan_reduce_workspace_floats": (c_sz, [c_int, c_int, c_int]),
    "uegan_instnorm_fwd": (c_int, [c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_f32, c_vp]),
    "uegan_instnorm_bwd": (c_int, [c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp]),
    "uegan_moments": (c_int, [c_int, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp]),
    "uegan_affine_act_fwd": (c_int, [c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp]),
    "uegan_affine_act_bwd_sums": (c_int, [c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp]),
    "uegan_affine_act_bwd_apply": (c_int, [c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp]),
    "uegan_rals_fwd": (c_int, [c_int, C.POINTER(c_vp), C.POINTER(c_vp), C.POINTER(c_i64), c_int, c_vp, c_vp, c_vp]),
    "uegan_rals_bwd": (c_int, [c_int, C.POINTER(c_vp), C.POINTER(c_vp), C.POINTER(c_i64), c_int, c_vp, c_vp, C.POINTER(c_vp), C.POINTER(c_vp), c_vp]),
    "uegan_pred_loss_fwd": (c_int, [c_int, c_f32, c_int, C.POINTER(c_vp), C.POINTER(c_i64), c_vp, c_vp, c_vp]),
    "uegan_pred_loss_bwd": (c_int, [c_int, c_f32, c_int, C.POINTER(c_vp), C.POINTER(c_i64), c_vp, C.POINTER(c_vp), c_vp]),
    "uegan_rahinge_fwd": (c_int, [c_int, C.POINTER(c_vp), C.POINTER(c_vp), C.POINTER(c_i64), c_int, c_vp, c_vp, c_vp]),
    "uegan_rahinge_bwd": (c_int, [c_int, C.POINTER(c_vp), C.POINTER(c_vp), C.POINTER(c_i64), c_int, c_vp, c_vp, C.POINTER(c_vp),
                                  C.POINTER(c_vp), c_vp]),
    "uegan_rahinge_heads_workspace_floats": (c_sz, [c_int]),
    "uegan_rahinge_heads_fwd": (c_int, [c_int, c_int, C.POINTER(c_vp), C.POINTER(c_i64), c_int, c_int, c_int, c_int, C.POINTER(C.c_int32), c_int,
                                        c_vp, c_vp, c_vp]),
    "uegan_rahinge_heads_bwd": (c_int, [c_int, c_int, C.POINTER(c_vp), C.POINTER(c_i64), c_int, c_int, c_int, c_int, C.POINTER(C.c_int32), c_int,
                                        c_vp, c_vp, C.POINTER(c_vp), C.c_uint32, c_vp]),
    "uegan_msrec_scratch_floats": (c_sz, []),
    "uegan_msrec_fwd": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "uegan_msrec_bwd": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "uegan_msl1_scratch_floats": (c_sz, []),
    "uegan_msl1_fwd": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "uegan_msl1_bwd": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "uegan_rahinge_workspace_floats": (c_sz, [c_int]),
    "uegan_pred_loss_workspace_floats": (c_sz, [c_int]),
    "uegan_specnorm_grad_workspace_floats": (c_sz, []),
    "uegan_sn_act_bwd_workspace_floats": (c_sz, [c_int, c_int]),
    "uegan_sn_act_bwd": (c_int, [c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_vp]),
    "uegan_sn_act_bwd_p": (c_int, [c_int, c_int, c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_int, c_int, c_vp]),
    "uegan_act_bwd_p": (c_int, [c_int, c_int, c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "uegan_sn_grad_finish": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "uegan_percep_tap_fwd": (c_int, [c_int, c_vp, c_vp, c_f32, c_vp, c_vp, c_int, c_int, c_int, c_f32, c_vp]),
    "uegan_percep_tap_fwd_given": (c_int, [c_int, c_vp, c_vp, c_f32, c_vp, c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "uegan_percep_tap_bwd": (c_int, [c_int, c_vp, c_vp, c_f32, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_f32, c_vp]),
    "uegan_percep_tap_bwd_act": (c_int, [c_int, c_int, c_vp, c_vp, c_f32, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_f32, c_vp]),
    "uegan_percep_tap_bwd_acc": (c_int, [c_int, c_int, c_vp, c_vp, c_f32, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_f32, c_int, c_vp]),
    "uegan_specnorm_sigma": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_f32, c_vp, c_vp, c_vp]),
    "uegan_specnorm_grad": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_vp, c_vp]),
    "uegan_copy_images": (c_int, [c_vp, c_vp, c_vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), c_int, c_i64, c_vp]),
    "uegan_quantize_u8": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "uegan_input_transform": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_vp, c_vp]),
    "uegan_image_metrics_u8": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "uegan_specnorm_multi_workspace_floats": (c_sz, [c_int, c_int]),
    "uegan_specnorm_multi": (c_int, [C.POINTER(SnLayer), c_int, c_int, c_int, c_f32, c_vp]),
    "uegan_specnorm_grad_acc": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_vp, c_int, c_vp]),
    "uegan_fill_zero": (c_int, [c_vp, c_sz, c_vp]),
    "uegan_scalar_wsum": (c_int, [c_int, C.POINTER(c_vp), C.POINTER(c_f32), c_vp, c_vp, c_vp]),
    "uegan_scalar_wsum_bwd": (c_int, [c_int, C.POINTER(c_f32), c_vp, c_vp, c_vp]),
    "uegan_gather_scalars": (c_int, [c_int, C.POINTER(c_vp), c_vp, c_vp]),
    "uegan_rmsprop_step": (c_int, [c_vp, c_int, c_i64, c_f32, c_f32, c_f32, c_f32, c_vp]),
    "uegan_adam_l2_step": (c_int, [c_vp, c_int, c_i64, c_f32, c_f32, c_f32, c_f32, c_f32, c_f32, c_int, c_vp]),
}


def _bind(cdll):
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(cdll, name)          # AttributeError here = ABI mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    return cdll


def load(path=None):
    """Load (once) and return the bound library (the fp16-format build while use_half_format("fp16") is in force). Raises if it has not been built."""
    global _lib, _lib_f16
    if _use_f16 and _emulated:
        return _emu_f16
    if _use_f16 and path is None:
        if _lib_f16 is None:
            if not os.path.exists(LIB_PATH_F16):
                raise RuntimeError(
                    "uegan_amd: %s not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                    "(hipcc --offload-arch=gfx950 -DUEGAN_HALF_FP16). There is no CPU fallback." % LIB_PATH_F16)
            _lib_f16 = _bind(C.CDLL(LIB_PATH_F16))
        return _lib_f16
    if _lib is None:
        path = path or LIB_PATH
        if not os.path.exists(path):
            raise RuntimeError(
                "uegan_amd: %s not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback." % path)
        _lib = _bind(C.CDLL(path))
    return _lib


def use_half_format(fmt):
    """which build serves 16-bit tensors from now on: "bf16" (libuegan_hip.so) or "fp16" (libuegan_hip_f16.so)"""
    global _use_f16
    if fmt not in ("bf16", "fp16"):
        raise ValueError(fmt)
    if fmt == "fp16" and _emulated and _emu_f16 is None:
        raise RuntimeError("no fp16-format emulator build was injected")
    _use_f16 = fmt == "fp16"


def is_emulated():
    return _emulated


def half_format():
    """the 16-bit storage format of the build that serves 16-bit tensors now: "bf16" or "fp16" (use_half_format)"""
    return "fp16" if _use_f16 else "bf16"


_emu_f16 = None


def _inject_for_tests(path, path_f16=None):
    """TESTS ONLY: bind the kernel sources compiled against tests/emu (CPU fiber emulator); path_f16: the fp16-format build of the same."""
    global _lib, _emulated, _emu_f16
    _lib = _bind(C.CDLL(path))
    _emu_f16 = _bind(C.CDLL(path_f16)) if path_f16 and os.path.exists(path_f16) else None
    _emulated = True
    return _lib


def _reset_for_tests():
    """TESTS ONLY: drop an injected emulator binding so the next load() binds the real library."""
    global _lib, _emulated, _emu_f16, _use_f16
    _lib = None
    _emu_f16 = None
    _emulated = False
    _use_f16 = False


n_calls = 0          # C-ABI calls checked so far (bench.py --gpus N reports calls per step: the host-side launch path of a rank)


def check(rc):
    global n_calls
    n_calls += 1
    if rc != 0:
        msg = load().uegan_last_error()
        raise RuntimeError("libuegan_hip error %d: %s" % (rc, msg.decode() if msg else "?"))
