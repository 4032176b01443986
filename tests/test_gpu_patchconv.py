"""Patch-staged 3x3 convolutions of the style network's end layers (csrc/patchconv.hip; Style_net.py:32-62 decoder's last conv 64 -> 3,
:64-118 encoder's first conv): against torch's convolution on the same rounded inputs, against the implicit GEMM they replace
(policy patch_conv = 0), in the 16-bit element type and in the f16x2 fp32-grade form; tile edges (reflection at all four borders,
several tiles per row and column, N > 1), bias and ReLU epilogues."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ref(x, w, b, relu):
    y = F.conv2d(F.pad(x.double(), (1, 1, 1, 1), mode="reflect"), w.double(), None if b is None else b.double())
    return (torch.relu(y) if relu else y).permute(0, 2, 3, 1)


@pytest.mark.parametrize("shape", [(2, 8, 64, 3), (1, 12, 128, 3), (3, 4, 64, 16), (1, 256, 256, 3)], ids=lambda s: "x".join(map(str, s)))
def test_co16_patch_conv_matches_torch_and_the_igemm(shape):
    """64 -> Co <= 16, fp32 output: 16-bit operands within accumulation-order noise of torch's conv on the rounded operands AND of the
    igemm path; f16x2 operands <= 2e-6 * max of torch's fp64 convolution."""
    from uda_poseestimation_amd import _hip, ops
    N, H, W, Co = shape
    g = torch.Generator().manual_seed(H * 7 + W + Co)
    x = torch.randn(N, 64, H, W, generator=g)
    w = torch.randn(Co, 64, 3, 3, generator=g) / 24
    b = torch.randn(Co, generator=g)
    d = ops.conv_desc(N, H, W, 64, Co, 3, 1, 1, reflect=True)
    d_ig = ops.with_policy(d, _hip.policy(patch_conv=0))
    for et in (torch.bfloat16, torch.float16):          # both builds of the library
        xr, wr = x.to(et).float(), w.to(et).float()
        xs = xr.permute(0, 2, 3, 1).contiguous().to(et).cuda()
        wp = ops.pack_weight(wr.cuda(), d, dtype=et)
        for relu in (False, True):
            y = ops.conv2d_fwd(xs, wp, d, bias=b.cuda(), relu=relu, out_f32=True)
            y_ig = ops.conv2d_fwd(xs, wp, d_ig, bias=b.cuda(), relu=relu, out_f32=True)
            ref = _ref(xr, wr, b, relu)
            scale = ref.abs().max().item()
            assert y.shape == (N, H, W, Co) and y.dtype == torch.float32
            e = (y.cpu().double() - ref).abs().max().item() / scale
            e_ig = (y - y_ig).abs().max().item() / scale
            assert e < 1e-5 and e_ig < 1e-5, (et, relu, e, e_ig)
    # f16x2
    xsp = ops.to_nhwc_split(x.cuda(), 64)
    wsp = ops.f32_to_split(w.cuda().permute(0, 2, 3, 1).contiguous())
    y = ops.conv2d_fwd(xsp, wsp, d, bias=b.cuda(), out_f32=True)
    y_ig = ops.conv2d_fwd(xsp, wsp, d_ig, bias=b.cuda(), out_f32=True)
    ref = _ref(x, w, b, False)
    scale = ref.abs().max().item()
    e = (y.cpu().double() - ref).abs().max().item() / scale
    e_ig = (y - y_ig).abs().max().item() / scale
    print(f"co16 patch conv {shape}: f16x2 err {e:.2e} * max vs fp64, {e_ig:.2e} vs the igemm")
    assert e < 2e-6 and e_ig < 2e-6


def test_patch_conv_dispatch_conditions():
    """Geometries outside the patch kernels' domain (W not a multiple of 64, zero padding, residual-free but 16-bit output, Ci != 64) still
    take the implicit GEMM and agree with torch."""
    from uda_poseestimation_amd import _hip, ops
    g = torch.Generator().manual_seed(3)
    et = torch.bfloat16
    for (N, H, W, Ci, Co, refl) in ((1, 8, 48, 64, 3, True), (1, 8, 64, 64, 3, False), (1, 6, 64, 64, 3, True), (1, 8, 64, 128, 3, True)):
        x = torch.randn(N, Ci, H, W, generator=g).to(et).float()
        w = (torch.randn(Co, Ci, 3, 3, generator=g) / 24).to(et).float()
        d = ops.conv_desc(N, H, W, Ci, Co, 3, 1, 1, reflect=refl)
        y = ops.conv2d_fwd(x.permute(0, 2, 3, 1).contiguous().to(et).cuda(), ops.pack_weight(w.cuda(), d, dtype=et), d, out_f32=True)
        ref = (F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), w) if refl else F.conv2d(x, w, padding=1)).permute(0, 2, 3, 1)
        assert (y.cpu() - ref).abs().max().item() < 1e-4 * ref.abs().max().item()


@pytest.mark.parametrize("shape", [(2, 8, 64), (1, 12, 128), (1, 256, 256)], ids=lambda s: "x".join(map(str, s)))
def test_ci8_patch_conv_matches_torch_and_the_igemm(shape):
    """3 (padded to 8) -> 64, 16-bit / split output with bias + ReLU: the encoder's first 3x3 convolution."""
    from uda_poseestimation_amd import _hip, ops
    N, H, W = shape
    g = torch.Generator().manual_seed(H + W)
    x = torch.randn(N, 3, H, W, generator=g)
    w = torch.randn(64, 3, 3, 3, generator=g) / 5
    b = torch.randn(64, generator=g)
    d = ops.conv_desc(N, H, W, 8, 64, 3, 1, 1, reflect=True)
    d_ig = ops.with_policy(d, _hip.policy(patch_conv=0))
    for et, tol in ((torch.bfloat16, 1.2e-2), (torch.float16, 2e-3)):
        xr, wr = x.to(et).float(), w.to(et).float()
        xs = torch.zeros(N, H, W, 8, dtype=et)
        xs[..., :3] = xr.permute(0, 2, 3, 1).to(et)
        xs = xs.cuda()
        wp = ops.pack_weight(wr.cuda(), d, dtype=et)
        for relu in (False, True):
            y = ops.conv2d_fwd(xs, wp, d, bias=b.cuda(), relu=relu)
            y_ig = ops.conv2d_fwd(xs, wp, d_ig, bias=b.cuda(), relu=relu)
            ref = _ref(xr, wr, b, relu)
            scale = ref.abs().max().item()
            assert y.shape == (N, H, W, 64) and y.dtype == et
            e = (y.cpu().double() - ref).abs().max().item() / scale
            assert e < tol, (et, relu, e)
            assert (y.float() - y_ig.float()).abs().max().item() <= tol * scale          # (both round the same fp32 sums; order differs)
    # f16x2
    xsp = ops.to_nhwc_split(x.cuda(), 8)
    wz = torch.zeros(64, 3, ops.kwp(d), 8, device="cuda")
    wz[:, :, :3, :3] = w.cuda().permute(0, 2, 3, 1)
    wsp = ops.f32_to_split(wz.contiguous())
    for relu in (False, True):
        y = ops.split_to_f32(ops.conv2d_fwd(xsp, wsp, d, bias=b.cuda(), relu=relu))
        y_ig = ops.split_to_f32(ops.conv2d_fwd(xsp, wsp, d_ig, bias=b.cuda(), relu=relu))
        ref = _ref(x, w, b, relu)
        scale = ref.abs().max().item()
        e = (y.cpu().double() - ref).abs().max().item() / scale
        e_ig = (y - y_ig).abs().max().item() / scale
        assert e < 2e-6 and e_ig < 2e-6, (relu, e, e_ig)
    print(f"ci8 patch conv {shape}: f16x2 err {e:.2e} * max vs fp64, {e_ig:.2e} vs the igemm")


TRUNK = [
    # N, H, W (input), Ci, Co, upsample
    (2, 8, 64, 64, 64, False),
    (1, 4, 64, 128, 128, False),
    (2, 32, 32, 256, 128, False),          # 32-wide maps: the 4 x 32 tile
    (1, 16, 32, 128, 64, True),            # folded nearest x2 upsample: 32 x 64 output
    (1, 64, 64, 64, 64, True),             # 128 x 128 output, several tiles per row and column
    (1, 6, 128, 64, 192, False),
]


@pytest.mark.parametrize("case,mode", [(c, 2) for c in TRUNK] + [(TRUNK[1], 3), (TRUNK[2], 3)], ids=lambda v: "x".join(map(str, v)) if isinstance(v, tuple) else f"mode{v}")
def test_trunk_patch_conv_matches_torch_and_the_igemm(case, mode):
    """patch3x3_kernel (policy patch_conv = 2; 3: 128 output channels per work-group in the 16-bit form where Co % 128 == 0): the style
    network's trunk layers, 16-bit and f16x2, bias + ReLU, with and without the folded upsample - against torch's convolution on the same
    rounded operands and against the implicit GEMM."""
    from uda_poseestimation_amd import _hip, ops
    N, H, W, Ci, Co, up = case
    g = torch.Generator().manual_seed(H * 3 + W + Ci + Co)
    x = torch.randn(N, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, 3, 3, generator=g) / np.sqrt(9 * Ci)
    b = torch.randn(Co, generator=g)
    d = ops.conv_desc(N, H, W, Ci, Co, 3, 1, 1, reflect=True, upsample=up, policy=_hip.policy(patch_conv=mode))
    d_ig = ops.with_policy(d, _hip.policy(patch_conv=0))

    def ref_of(xr, wr, relu):
        xu = F.interpolate(xr, scale_factor=2, mode="nearest") if up else xr
        return _ref(xu, wr, b, relu)
    for et, tol in ((torch.bfloat16, 1.2e-2), (torch.float16, 2e-3)):
        xr, wr = x.to(et).float(), w.to(et).float()
        xs = xr.permute(0, 2, 3, 1).contiguous().to(et).cuda()
        wp = ops.pack_weight(wr.cuda(), d, dtype=et)
        for relu in (False, True):
            y = ops.conv2d_fwd(xs, wp, d, bias=b.cuda(), relu=relu)
            y_ig = ops.conv2d_fwd(xs, wp, d_ig, bias=b.cuda(), relu=relu)
            ref = ref_of(xr, wr, relu)
            scale = ref.abs().max().item()
            assert tuple(y.shape) == tuple(ref.shape) and y.dtype == et
            e = (y.cpu().double() - ref).abs().max().item() / scale
            assert e < tol, (et, relu, e)
            assert (y.float() - y_ig.float()).abs().max().item() <= tol * scale
    xsp = ops.to_nhwc_split(x.cuda(), Ci)
    wsp = ops.f32_to_split(w.cuda().permute(0, 2, 3, 1).contiguous())
    for relu in (False, True):
        y = ops.split_to_f32(ops.conv2d_fwd(xsp, wsp, d, bias=b.cuda(), relu=relu))
        y_ig = ops.split_to_f32(ops.conv2d_fwd(xsp, wsp, d_ig, bias=b.cuda(), relu=relu))
        ref = ref_of(x, w, relu)
        scale = ref.abs().max().item()
        e = (y.cpu().double() - ref).abs().max().item() / scale
        e_ig = (y - y_ig).abs().max().item() / scale
        assert e < 2e-6 and e_ig < 2e-6, (relu, e, e_ig)
    print(f"trunk patch conv {case}: f16x2 err {e:.2e} * max vs fp64, {e_ig:.2e} vs the igemm")
