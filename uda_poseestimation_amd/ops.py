"""Thin tensor-level wrappers over the C ABI (one function per kernel family).

Tensors are torch CUDA tensors used only as device memory; all arithmetic happens in libudapose_hip.so.
Activations: NHWC bf16 contiguous ([N,H,W,C]); weights packed bf16 [Co][taps][Ci]; heat-maps NCHW fp32.
"""
import ctypes as C

import torch

from . import _hip
from ._hip import ConvDesc, check, lib, lib_for, ptr, require_cuda, stream

EPI_RELU, EPI_OUT_F32, EPI_F32, EPI_SPLIT = 1, 2, 4, 8
# "f16x2" split tensors (include/udapose.h, UDAPOSE_EPI_SPLIT: the fast fp32-grade mode) have the byte footprint of the fp32 tensor
# of the same shape; torch sees them as int32 tensors (never interpreted by torch: only handed to the library)
SPLIT = torch.int32


def conv_desc(N, Hi, Wi, Ci, Co, K, stride=1, pad=0, transposed=False, reflect=False, upsample=False, policy=None):
    """`policy`: a _hip.Policy (explicit dispatch policy of calls made with this descriptor), None = production policy."""
    d = ConvDesc(N, Hi, Wi, Ci, Co, K, K, stride, pad, int(transposed), int(reflect), int(upsample))
    if policy is not None:
        d.policy = C.pointer(policy)        # (ctypes keeps the Policy object alive with the descriptor)
    return d


def with_policy(d, policy):
    """A copy of descriptor d that names another dispatch policy."""
    return conv_desc(d.N, d.Hi, d.Wi, d.Ci, d.Co, d.KH, d.stride, d.pad, d.transposed, d.reflect, d.upsample, policy)


def conv_out_hw(d):
    ho, wo = C.c_int(), C.c_int()
    lib().udapose_conv_out_hw(C.byref(d), C.byref(ho), C.byref(wo))
    return ho.value, wo.value


def kwp(d):
    return (d.KW + 7) // 8 * 8 if d.Ci == 8 else d.KW


def pack_weight(w, d, direction="fwd", dtype=torch.bfloat16):
    """fp32 torch-layout weight ([Co,Ci,KH,KW]; transposed: [Ci,Co,KH,KW]) -> packed bf16 (or fp16) for fprop or dgrad."""
    require_cuda(w)
    L = lib("fp16" if dtype == torch.float16 else "bf16")
    T = d.KH * d.KW
    wc = w.detach().float().contiguous(memory_format=torch.channels_last)   # physical [A][KH][KW][B]
    flat = wc.permute(0, 2, 3, 1)                                           # logical [A,KH,KW,B], contiguous
    A, B = flat.shape[0], flat.shape[3]
    if d.Ci == 8:
        assert direction == "fwd" and not d.transposed
        out = torch.empty(d.Co, d.KH, kwp(d), 8, dtype=dtype, device=w.device)
        check(L.udapose_pack_strided(stream(), ptr(flat), ptr(out), d.Co, d.KH, kwp(d), d.KW, 8, B, d.KH * d.KW * B, d.KW * B, B, 1), "pack")
        return out
    direct = (direction == "fwd") != bool(d.transposed)
    if direct:
        out = torch.empty(A, T, B, dtype=dtype, device=w.device)
        check(L.udapose_cast_f32_bf16(stream(), ptr(flat), ptr(out), flat.numel()), "cast")
    else:
        out = torch.empty(B, T, A, dtype=dtype, device=w.device)
        check(L.udapose_transpose_cast(stream(), ptr(flat), ptr(out), A, T, B), "transpose_cast")
    return out


def conv2d_fwd(x, w_fwd, d, res=None, bias=None, relu=False, out_f32=False, want_stats=False):
    require_cuda(x, w_fwd)
    f32 = x.dtype == torch.float32          # exact fp32 path: w_fwd must be fp32 [Co][taps][Ci] too
    split = x.dtype == SPLIT                # f16x2 path: x, w_fwd (and y unless out_f32) are split tensors
    assert x.dtype in (torch.bfloat16, torch.float16, torch.float32, SPLIT) and w_fwd.dtype == x.dtype and x.is_contiguous()
    assert tuple(x.shape) == (d.N, d.Hi, d.Wi, d.Ci)
    ho, wo = conv_out_hw(d)
    y = torch.empty(d.N, ho, wo, d.Co, dtype=torch.float32 if (out_f32 or f32) else x.dtype, device=x.device)
    stats = None
    if want_stats:
        rows = lib().udapose_conv_stat_rows(C.byref(d))
        stats = torch.empty(rows, 2, d.Co, dtype=torch.float32, device=x.device)
    flags = (EPI_RELU if relu else 0) | (EPI_OUT_F32 if out_f32 else 0) | (EPI_F32 if f32 else 0) | (EPI_SPLIT if split else 0)
    check(lib_for(x).udapose_conv2d_fwd(stream(), C.byref(d), ptr(x), ptr(w_fwd), ptr(y), ptr(res), ptr(bias), ptr(stats), flags), "conv2d_fwd")
    return (y, stats) if want_stats else y


def conv2d_bwd_data(dy, w_bwd, d, res=None, out_f32=False):
    require_cuda(dy, w_bwd)
    dx = torch.empty(d.N, d.Hi, d.Wi, d.Ci, dtype=torch.float32 if out_f32 else dy.dtype, device=dy.device)
    check(lib_for(dy).udapose_conv2d_bwd_data(stream(), C.byref(d), ptr(dy), ptr(w_bwd), ptr(dx), ptr(res), int(out_f32)), "conv2d_bwd_data")
    return dx


def conv2d_bwd_data_bn(dy, w_bwd, d, bn_y, bn_mean, bn_invstd, bn_z=None, bn_gamma=None, bn_beta=None, res=None, out_f32=False):
    """dgrad whose output feeds a training-mode BatchNorm (+ReLU) backward: returns (g, slab) with g = dz * relu-mask and
    slab[rows][2][Ci] the per-m-tile partial sums of g and g * xhat (udapose_conv2d_bwd_data_bn)."""
    require_cuda(dy, w_bwd, bn_y, bn_mean, bn_invstd)
    assert tuple(bn_y.shape) == (d.N, d.Hi, d.Wi, d.Ci) and bn_y.dtype == dy.dtype and bn_y.is_contiguous()
    assert bn_z is not None or (bn_gamma is not None and bn_beta is not None)
    dx = torch.empty(d.N, d.Hi, d.Wi, d.Ci, dtype=torch.float32 if out_f32 else dy.dtype, device=dy.device)
    rows = lib().udapose_conv_bwd_stat_rows(C.byref(d))
    if rows < 1:
        raise RuntimeError(f"conv_bwd_stat_rows: {rows}")
    slab = torch.empty(rows, 2, d.Ci, dtype=torch.float32, device=dy.device)
    check(lib_for(dy).udapose_conv2d_bwd_data_bn(stream(), C.byref(d), ptr(dy), ptr(w_bwd), ptr(dx), ptr(res), int(out_f32), ptr(bn_y), ptr(bn_z),
                                           ptr(bn_mean), ptr(bn_invstd), ptr(bn_gamma), ptr(bn_beta), ptr(slab)), "conv2d_bwd_data_bn")
    return dx, slab


def conv2d_bwd_weight(dy, x, d, dw=None):
    require_cuda(dy, x)
    T = d.KH * kwp(d)
    shape = (d.Ci, T, d.Co) if d.transposed else (d.Co, T, d.Ci)
    acc = dw is not None
    if dw is None:
        dw = torch.empty(shape, dtype=torch.float32, device=x.device)
    check(lib_for(x).udapose_conv2d_bwd_weight(stream(), C.byref(d), ptr(dy), ptr(x), ptr(dw), int(acc)), "conv2d_bwd_weight")
    return dw


def to_nhwc_bf16(x_nchw, cpad=None, dtype=torch.bfloat16):
    """NCHW fp32 -> NHWC in the 16-bit element type (bf16 by default; torch.float16 uses the fp16 build)."""
    require_cuda(x_nchw)
    N, Cc, H, W = x_nchw.shape
    cpad = cpad or (Cc + 7) // 8 * 8
    out = torch.empty(N, H, W, cpad, dtype=dtype, device=x_nchw.device)
    check(lib_for(out).udapose_nchw_f32_to_nhwc_bf16(stream(), ptr(x_nchw.float().contiguous()), ptr(out), N, Cc, H * W, cpad), "to_nhwc")
    return out


def to_nhwc_f32(x_nchw, cpad=None):
    """NCHW fp32 -> NHWC fp32 (channels zero-padded to cpad): the style path at the reference's precision."""
    require_cuda(x_nchw)
    N, Cc, H, W = x_nchw.shape
    cpad = cpad or (Cc + 7) // 8 * 8
    out = torch.empty(N, H, W, cpad, dtype=torch.float32, device=x_nchw.device)
    check(lib().udapose_nchw_f32_to_nhwc_f32(stream(), ptr(x_nchw.float().contiguous()), ptr(out), N, Cc, H * W, cpad), "to_nhwc_f32")
    return out


def to_nhwc_split(x_nchw, cpad=None):
    """NCHW fp32 -> NHWC f16x2 split (channels zero-padded to cpad): the style path in the fast fp32-grade mode."""
    require_cuda(x_nchw)
    N, Cc, H, W = x_nchw.shape
    cpad = cpad or (Cc + 7) // 8 * 8
    out = torch.empty(N, H, W, cpad, dtype=SPLIT, device=x_nchw.device)
    check(lib().udapose_nchw_f32_to_nhwc_split(stream(), ptr(x_nchw.float().contiguous()), ptr(out), N, Cc, H * W, cpad), "to_nhwc_split")
    return out


def f32_to_split(x):
    """fp32 tensor (contiguous, last dimension a multiple of 8) -> f16x2 split tensor of the same shape."""
    require_cuda(x)
    x = x.float().contiguous()
    assert x.shape[-1] % 8 == 0
    out = torch.empty(x.shape, dtype=SPLIT, device=x.device)
    check(lib().udapose_f32_to_split(stream(), ptr(x), ptr(out), x.numel()), "f32_to_split")
    return out


def split_to_f32(x):
    require_cuda(x)
    assert x.dtype == SPLIT and x.is_contiguous() and x.shape[-1] % 8 == 0
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    check(lib().udapose_split_to_f32(stream(), ptr(x), ptr(out), x.numel()), "split_to_f32")
    return out


def to_nchw_f32(x_nhwc, channels=None, lo=None, hi=None):
    require_cuda(x_nhwc)
    N, H, W, Cs = x_nhwc.shape
    Cc = channels or Cs
    out = torch.empty(N, Cc, H, W, dtype=torch.float32, device=x_nhwc.device)
    kind = 2 if x_nhwc.dtype == SPLIT else int(x_nhwc.dtype == torch.float32)
    check(lib_for(x_nhwc).udapose_nhwc_to_nchw_f32(stream(), ptr(x_nhwc), kind, ptr(out), N, Cc, H * W, Cs, ptr(lo), ptr(hi)),
          "to_nchw")
    return out


def bn_train_fwd(y, stats, gamma, beta, running_mean, running_var, nbt, momentum=0.1, eps=1e-5, res=None, relu=True):
    C_ = y.shape[-1]
    dev = y.device
    scale, shift, mean, invstd = (torch.empty(C_, dtype=torch.float32, device=dev) for _ in range(4))
    count = float(y.numel() // C_)
    check(lib().udapose_bn_finalize(stream(), ptr(stats), stats.shape[0], C_, count, ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var),
                                    ptr(nbt), momentum, eps, ptr(scale), ptr(shift), ptr(mean), ptr(invstd)), "bn_finalize")
    z = torch.empty_like(y)
    check(lib_for(y).udapose_bn_apply(stream(), ptr(y), ptr(res), ptr(z), y.numel(), C_, ptr(scale), ptr(shift), int(relu)), "bn_apply")
    return z, mean, invstd


def bn_bwd(dz, z, y, gamma, mean, invstd, relu=True, want_g=False, beta=None):
    """relu: False/0 none, True/1 mask from z, 2 mask recomputed from y (needs beta, z may be None)."""
    C_ = y.shape[-1]
    npix = y.numel() // C_
    dev = y.device
    rows = lib().udapose_bn_bwd_rows(npix)
    slab = torch.empty(rows, 2, C_, dtype=torch.float32, device=dev)
    coef = torch.empty(3, C_, dtype=torch.float32, device=dev)
    dgamma = torch.empty(C_, dtype=torch.float32, device=dev)
    dbeta = torch.empty(C_, dtype=torch.float32, device=dev)
    dy = torch.empty_like(y)
    g = torch.empty_like(y) if want_g else None
    check(lib_for(y).udapose_bn_bwd(stream(), ptr(dz), int(dz.dtype == torch.float32), ptr(z), ptr(y), ptr(dy), ptr(g), npix, C_, ptr(gamma), ptr(mean),
                               ptr(invstd), int(relu),
                               ptr(slab), ptr(coef), ptr(dgamma), ptr(dbeta), 0.0, ptr(beta)), "bn_bwd")
    return dy, dgamma, dbeta, g


def bn_bwd_pre(g, y, gamma, mean, invstd, slab):
    """BN backward from an already masked, already reduced gradient (conv2d_bwd_data_bn's outputs): (dy, dgamma, dbeta)."""
    C_ = y.shape[-1]
    npix = y.numel() // C_
    dev = y.device
    coef = torch.empty(3, C_, dtype=torch.float32, device=dev)
    dgamma = torch.empty(C_, dtype=torch.float32, device=dev)
    dbeta = torch.empty(C_, dtype=torch.float32, device=dev)
    dy = torch.empty_like(y)
    check(lib_for(y).udapose_bn_bwd_pre(stream(), ptr(g), int(g.dtype == torch.float32), ptr(y), ptr(dy), npix, C_, ptr(gamma), ptr(mean), ptr(invstd),
                                   ptr(slab), slab.shape[0], ptr(coef), ptr(dgamma), ptr(dbeta), 0.0), "bn_bwd_pre")
    return dy, dgamma, dbeta


def maxpool3x3s2_fwd(x):
    N, H, W, C_ = x.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.empty(N, Ho, Wo, C_, dtype=x.dtype, device=x.device)
    idx = torch.empty(N, Ho, Wo, C_, dtype=torch.uint8, device=x.device)
    check(lib_for(x).udapose_maxpool3x3s2_fwd(stream(), ptr(x), ptr(y), ptr(idx), N, H, W, C_), "maxpool_fwd")
    return y, idx


def maxpool3x3s2_bwd(dy, idx, H, W):
    N, _, _, C_ = dy.shape
    dx = torch.empty(N, H, W, C_, dtype=dy.dtype, device=dy.device)
    check(lib_for(dy).udapose_maxpool3x3s2_bwd(stream(), ptr(dy), ptr(idx), ptr(dx), N, H, W, C_), "maxpool_bwd")
    return dx


def maxpool2x2_ceil(x):
    N, H, W, C_ = x.shape
    y = torch.empty(N, (H + 1) // 2, (W + 1) // 2, C_, dtype=x.dtype, device=x.device)
    fn = lib().udapose_maxpool2x2_ceil_split if x.dtype == SPLIT else (
        lib().udapose_maxpool2x2_ceil_f32 if x.dtype == torch.float32 else lib_for(x).udapose_maxpool2x2_ceil)
    check(fn(stream(), ptr(x), ptr(y), N, H, W, C_), "maxpool2x2")
    return y


def adain(content, style, alpha=1.0, eps=1e-5, want_stats=False, stats_only=False):
    """NHWC bf16 or fp32 (both operands alike).  stats_only: no output tensor, [N,C,4] = (mean_c, std_c, mean_s, std_s)."""
    N, H, W, C_ = content.shape
    assert content.dtype == style.dtype and content.dtype in (torch.bfloat16, torch.float16, torch.float32, SPLIT)
    assert style.shape[0] == N and style.shape[3] == C_
    out = None if stats_only else torch.empty_like(content)
    st = torch.empty(N, C_, 4, dtype=torch.float32, device=content.device) if (want_stats or stats_only) else None
    f32 = content.dtype == torch.float32
    if content.dtype == SPLIT:
        dev_alpha = torch.is_tensor(alpha)
        if dev_alpha:
            assert alpha.is_cuda and alpha.dtype == torch.float32 and alpha.numel() == 1
        check(lib().udapose_adain_split(stream(), ptr(content), ptr(style), ptr(out), N, H * W, style.shape[1] * style.shape[2], C_, eps,
                                        1.0 if dev_alpha else float(alpha), ptr(alpha) if dev_alpha else None, ptr(st)), "adain")
    elif torch.is_tensor(alpha):      # the blend factor as ONE fp32 device scalar: read at run time (captured launches follow it)
        assert alpha.is_cuda and alpha.dtype == torch.float32 and alpha.numel() == 1
        L = lib() if f32 else lib_for(content)
        check(L.udapose_adain_alpha_dev(stream(), ptr(content), ptr(style), ptr(out), N, H * W, style.shape[1] * style.shape[2], C_, eps,
                                        ptr(alpha), ptr(st), int(f32)), "adain")
    else:
        fn = lib().udapose_adain_f32 if f32 else lib_for(content).udapose_adain
        check(fn(stream(), ptr(content), ptr(style), ptr(out), N, H * W, style.shape[1] * style.shape[2], C_, eps, float(alpha), ptr(st)), "adain")
    if stats_only:
        return st
    return (out, st) if want_stats else out
