"""ctypes binding of libudapose_hip.so (C ABI declared in include/udapose.h).

The product path has no CPU or eager-PyTorch fallback: if the library is missing or a call fails, we raise.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libudapose_hip.so")
# the same sources built twice (csrc/Makefile): element type bf16 (the benched precision) and fp16 (the reference's autocast dtype)
LIB_PATHS = {"bf16": LIB_PATH, "fp16": os.path.join(_HERE, "libudapose_hip_f16.so")}
_libs = {}

vp, ci, cf, cd, sz, ll, cl = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_size_t, C.c_longlong, C.c_long


class Policy(C.Structure):
    """udapose_policy: the explicit dispatch policy (include/udapose.h).  `policy()` returns the production policy with the
    given fields overridden; tests force code paths with it, bench.py's tuning flags run A/B comparisons through it."""
    _fields_ = [(k, C.c_int) for k in ("igemm_tile", "igemm_h3", "igemm_lean", "igemm_short_lds", "igemm_tap0", "wgrad_tile", "wgrad_ksplit",
                                       "wgrad_fastgeo", "wgrad_group", "wgrad_stages", "wgrad_group_stem", "bn_bwd_fused", "bn_fwd_chunked",
                                       "bn_bwd_chunked", "bn_bwd_pre_legacy", "igemm_wg_min", "wgrad_row3", "bn3_mask", "stem_fused", "debug_sync", "igemm_big_min", "patch_conv",
                                       "eval_fold", "bn_xcd_rows", "wgrad_det", "igemm_ns3_k")] + [("timeline", C.c_void_p)]


def policy(**overrides):
    p = Policy()
    lib().udapose_policy_default(C.byref(p))
    for k, v in overrides.items():
        if k not in dict(Policy._fields_):
            raise KeyError(f"udapose_policy has no field {k!r}")
        setattr(p, k, v)
    return p


class ConvDesc(C.Structure):
    _fields_ = [(k, C.c_int) for k in ("N", "Hi", "Wi", "Ci", "Co", "KH", "KW", "stride", "pad", "transposed", "reflect", "upsample")] + \
               [("policy", C.POINTER(Policy))]


_SIGS = {
    "udapose_version": (ci, []),
    "udapose_elem_kind": (ci, []),
    "udapose_grad_scaler_check": (ci, [vp, vp, vp, vp, vp, ci, vp]),
    "udapose_grad_scaler_check2": (ci, [vp, vp, vp, vp, vp, ci, vp, ll]),
    "udapose_grad_scaler_update": (ci, [vp, vp, cf, cf, ci]),
    "udapose_policy_default": (None, [vp]),
    "udapose_conv_prepare": (ci, [vp]),
    "udapose_net_set_policy": (ci, [vp, vp]),
    "udapose_net_get_policy": (ci, [vp, vp]),
    "udapose_net_bind": (ci, [vp, vp, vp, vp]),
    "udapose_net_bind_grads": (ci, [vp, vp]),
    "udapose_conv_out_hw": (None, [vp, vp, vp]),
    "udapose_conv_stat_rows": (ci, [vp]),
    "udapose_conv2d_fwd": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, ci]),
    "udapose_conv2d_bwd_data": (ci, [vp, vp, vp, vp, vp, vp, ci]),
    "udapose_conv_bwd_stat_rows": (ci, [vp]),
    "udapose_conv2d_bwd_data_bn": (ci, [vp, vp, vp, vp, vp, vp, ci, vp, vp, vp, vp, vp, vp, vp]),
    "udapose_conv2d_bwd_weight": (ci, [vp, vp, vp, vp, vp, ci]),
    "udapose_cast_f32_bf16": (ci, [vp, vp, vp, sz]),
    "udapose_transpose_cast": (ci, [vp, vp, vp, ci, ci, ci]),
    "udapose_pack_strided": (ci, [vp, vp, vp, ci, ci, ci, ci, ci, ci, cl, cl, cl, cl]),
    "udapose_nchw_f32_to_nhwc_bf16": (ci, [vp, vp, vp, ci, ci, ci, ci]),
    "udapose_nchw_f32_to_nhwc_f32": (ci, [vp, vp, vp, ci, ci, ci, ci]),
    "udapose_nhwc_to_nchw_f32": (ci, [vp, vp, ci, vp, ci, ci, ci, ci, vp, vp]),
    "udapose_nchw_f32_to_nhwc_split": (ci, [vp, vp, vp, ci, ci, ci, ci]),
    "udapose_f32_to_split": (ci, [vp, vp, vp, sz]),
    "udapose_split_to_f32": (ci, [vp, vp, vp, sz]),
    "udapose_maxpool2x2_ceil_split": (ci, [vp, vp, vp, ci, ci, ci, ci]),
    "udapose_adain_split": (ci, [vp, vp, vp, vp, ci, ci, ci, ci, cf, cf, vp, vp]),
    "udapose_bn_finalize": (ci, [vp, vp, ci, ci, cd, vp, vp, vp, vp, vp, cf, cf, vp, vp, vp, vp]),
    "udapose_bn_eval_coeff": (ci, [vp, ci, vp, vp, vp, vp, cf, vp, vp]),
    "udapose_bn_apply": (ci, [vp, vp, vp, vp, sz, ci, vp, vp, ci]),
    "udapose_bn_bwd_rows": (ci, [sz]),
    "udapose_bn_bwd": (ci, [vp, vp, ci, vp, vp, vp, vp, sz, ci, vp, vp, vp, ci, vp, vp, vp, vp, cf, vp]),
    "udapose_bn_bwd_pre": (ci, [vp, vp, ci, vp, vp, sz, ci, vp, vp, vp, vp, ci, vp, vp, vp, cf]),
    "udapose_maxpool3x3s2_fwd": (ci, [vp, vp, vp, vp, ci, ci, ci, ci]),
    "udapose_maxpool3x3s2_bwd": (ci, [vp, vp, vp, vp, ci, ci, ci, ci]),
    "udapose_maxpool2x2_ceil": (ci, [vp, vp, vp, ci, ci, ci, ci]),
    "udapose_maxpool2x2_ceil_f32": (ci, [vp, vp, vp, ci, ci, ci, ci]),
    "udapose_net_create": (ci, [vp, ci, ci, ci, ci, ci, vp]),
    "udapose_net_destroy": (None, [vp]),
    "udapose_net_num_params": (ci, [vp]),
    "udapose_net_num_buffers": (ci, [vp]),
    "udapose_net_param_numel": (ll, [vp, ci]),
    "udapose_net_wpack_bytes": (sz, [vp]),
    "udapose_net_act_bytes": (sz, [vp]),
    "udapose_net_ws_bytes": (sz, [vp]),
    "udapose_net_out_shape": (None, [vp, vp]),
    "udapose_net_pack_weights": (ci, [vp, vp, vp, vp, ci]),
    "udapose_net_forward": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, cf]),
    "udapose_net_apply_running": (ci, [vp, vp, vp, vp, cf]),
    "udapose_axpy_f32": (ci, [vp, vp, vp, sz]),
    "udapose_net_backward": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, cf]),
    "udapose_net_backward_part": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, cf, ci]),
    "udapose_net_backward_phase": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, cf, ci, ci]),
    "udapose_net_wgrad_pair": (ci, [vp, vp, vp, vp, vp, cf, vp, vp, vp, cf, ci]),
    "udapose_net_grad_split_param": (ll, [vp]),
    "udapose_net_bind_update": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "udapose_net_fused_update": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, vp, cf, cf, cf, cf, cf, ci, cf, vp, cf, cf, ci, ll]),
    "udapose_joints_mse_fwd": (ci, [vp, vp, vp, vp, ci, ci, vp, vp]),
    "udapose_joints_mse_bwd": (ci, [vp, vp, vp, vp, vp, ci, ci, vp]),
    "udapose_cons_loss_fwd": (ci, [vp, vp, vp, vp, ci, ci, vp, vp]),
    "udapose_cons_loss_bwd": (ci, [vp, vp, vp, vp, vp, ci, ci, vp]),
    "udapose_mask_count": (ci, [vp, vp, sz, vp]),
    "udapose_cons_loss_valid_fwd": (ci, [vp, vp, vp, vp, vp, vp, ci, ci, ci, vp, vp]),
    "udapose_cons_loss_valid_bwd": (ci, [vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, vp]),
    "udapose_heatmap_argmax": (ci, [vp, vp, ci, ci, ci, vp, vp, vp, vp, vp, ci]),
    "udapose_kth_mask": (ci, [vp, vp, vp, ci, ci, vp, vp, vp, ci]),
    "udapose_pck": (ci, [vp, vp, vp, ci, ci, cf, cf, cf, vp, vp]),
    "udapose_multi_chunk": (ci, []),
    "udapose_ema_multi": (ci, [vp, vp, vp, vp, vp, vp, ci, cf, cf]),
    "udapose_adam_multi": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, ci, cf, cf, cf, cf, cf, ci, cf, vp]),
    "udapose_sgd_multi": (ci, [vp, vp, vp, vp, vp, vp, vp, ci, cf, cf, cf, ci, ci, cf, vp]),
    "udapose_comm_pack_bf16": (ci, [vp, vp, ll, vp, ll]),
    "udapose_comm_shard_mean": (ci, [vp, vp, ci, ll, vp]),
    "udapose_comm_unpack_bf16": (ci, [vp, vp, vp, ll]),
    "udapose_adain": (ci, [vp, vp, vp, vp, ci, ci, ci, ci, cf, cf, vp]),
    "udapose_adain_f32": (ci, [vp, vp, vp, vp, ci, ci, ci, ci, cf, cf, vp]),
    "udapose_adain_alpha_dev": (ci, [vp, vp, vp, vp, ci, ci, ci, ci, cf, vp, vp, ci]),
    "udapose_patch_paste": (ci, [vp, vp, vp, ci, ci, ci, ci, ci]),
    "udapose_occlusion_pick": (ci, [vp, vp, vp, vp, ci, ci, ci, cd, ci, cf, cf, ci, vp, vp]),
    "udapose_select_rows": (ci, [vp, vp, vp, vp, vp, ci, sz]),
    "udapose_aug_affine_u8": (ci, [vp, vp, vp, vp, ci, ci, ci]),
    "udapose_aug_color_op": (ci, [vp, vp, vp, vp, vp, ci, ci]),
    "udapose_aug_to_tensor": (ci, [vp, vp, vp, ci, ci, vp, vp]),
    "udapose_aug_gaussian_blur_u8": (ci, [vp, vp, vp, vp, ci, ci, ci]),
    "udapose_aug_resized_crop_u8": (ci, [vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci]),
    "udapose_gaussian_labels": (ci, [vp, vp, vp, vp, vp, ci, ci, ci, cd, cd, vp, ci]),
    "udapose_split_saturations": (ci, [ci, vp]),
    "udapose_mean_views": (ci, [vp, vp, ci, vp, sz]),
    "udapose_draw_labelmap_ori": (ci, [vp, vp, vp, vp, vp, vp, ci, ci, ci, cf, vp, ci]),
    "udapose_prof_begin": (None, []),
    "udapose_prof_end": (ci, [vp]),
    "udapose_affine_nearest": (ci, [vp, vp, vp, vp, ci, ci, ci, ci, ci, ci]),
    "udapose_recon_thetas": (ci, [vp, vp, ci, cd, vp, vp]),
}
EXPORTS = tuple(_SIGS.keys())


class HipLibraryMissing(RuntimeError):
    pass


def lib(kind="bf16"):
    """Load a shared library (raises loudly if it has not been built: there is no fallback path).  kind: 'bf16' (default) or
    'fp16' = the element type of activations / packed weights / MFMA operands; everything else (losses, heat-map kernels,
    optimizers, warps) is fp32 and identical in both builds."""
    l = _libs.get(kind)
    if l is None:
        path = LIB_PATHS[kind]
        if not os.path.exists(path):
            raise HipLibraryMissing(
                f"{path} not found: build it with `make -C uda_poseestimation_amd/csrc` (or __graft_entry__.build()). "
                "uda_poseestimation_amd has no CPU / eager fallback.")
        l = C.CDLL(path)
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        if l.udapose_elem_kind() != {"bf16": 0, "fp16": 1}[kind]:
            raise HipLibraryMissing(f"{path} was built for another element type")
        _libs[kind] = l
    return l


def lib_for(*tensors):
    """The build whose element type matches the 16-bit tensors given (fp16 -> the fp16 build, otherwise bf16)."""
    for t in tensors:
        if t is not None and torch.is_tensor(t) and t.dtype == torch.float16:
            return lib("fp16")
    return lib("bf16")


ELEM_DTYPE = {"bf16": torch.bfloat16, "fp16": torch.float16}


_ERRORS = {-1: "bad argument", -2: "launch / runtime failure", -3: "unsupported configuration",
           -4: "not prepared: a device table was not built for these pointers (udapose_net_bind / udapose_net_bind_grads / udapose_conv_prepare)"}


def check(code, what=""):
    if code != 0:
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            # (an exception that unwinds a stream capture with forked streams can take the runtime down before it is ever shown)
            import sys
            print(f"libudapose_hip call failed inside a stream capture ({what}): error {code} ({_ERRORS.get(code, 'unknown')})",
                  file=sys.stderr, flush=True)
        raise RuntimeError(f"libudapose_hip call failed ({what}): error {code} ({_ERRORS.get(code, 'unknown')})")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("uda_poseestimation_amd runs on MI355X only: got a CPU tensor (there is no CPU fallback; "
                               "the CPU restatement lives in oracle/ and is test-only)")
