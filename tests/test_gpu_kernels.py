"""GPU parity tests of each kernel family, through the C ABI, against the torch-CPU fp32 oracle of the same op evaluated on
the SAME bf16-rounded inputs (so the only differences are accumulation order and the final bf16 rounding of outputs).
Tolerances: bf16 outputs -> |err| <= 1.2e-2 * max|ref| (bf16 has 8 bits of mantissa: 3.9e-3 relative per rounding);
fp32 outputs -> 1e-4 relative to max|ref|.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    from uda_poseestimation_amd import _hip
    _hip.lib()
    return torch.device("cuda:0")


def bf(x):
    return x.to(torch.bfloat16).float()


def nhwc(x):      # NCHW fp32 cpu -> NHWC bf16 cuda
    return x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).cuda()


def nchw(y):      # NHWC cuda -> NCHW fp32 cpu
    return y.float().cpu().permute(0, 3, 1, 2).contiguous()


def close(got, ref, tol):
    scale = ref.abs().max().item() + 1e-12
    err = (got - ref).abs().max().item()
    assert err <= tol * scale, f"max err {err:.3e} vs scale {scale:.3e} (tol {tol})"


CONV_CASES = [
    # name, N, H, W, Ci, Co, K, stride, pad
    ("1x1_64_256", 2, 16, 16, 64, 256, 1, 1, 0),
    ("1x1_s2", 2, 16, 16, 256, 512, 1, 2, 0),
    ("3x3_s1", 2, 16, 16, 64, 64, 3, 1, 1),
    ("3x3_s2", 2, 16, 16, 128, 128, 3, 2, 1),
    ("3x3_ragged", 3, 12, 12, 64, 128, 3, 1, 1),
    ("3x3_odd", 1, 9, 7, 64, 64, 3, 2, 1),
    ("1x1_big", 4, 32, 32, 256, 1024, 1, 1, 0),
    ("1x1_head32", 2, 16, 16, 256, 32, 1, 1, 0),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_fwd_bwd(dev, case):
    from uda_poseestimation_amd import ops
    _, N, H, W, Ci, Co, K, s, p = case
    g = torch.Generator().manual_seed(1)
    x = bf(torch.randn(N, Ci, H, W, generator=g))
    w = bf(torch.randn(Co, Ci, K, K, generator=g) / (Ci * K * K) ** 0.5)
    d = ops.conv_desc(N, H, W, Ci, Co, K, s, p)
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    ref = F.conv2d(xr, wr, stride=s, padding=p)
    y, stats = ops.conv2d_fwd(nhwc(x), ops.pack_weight(w.cuda(), d, "fwd"), d, want_stats=True)
    close(nchw(y), ref.detach(), 1.2e-2)
    # fused BN statistics: column sums of the fp32 result
    ssum = stats.double().sum(0).cpu()
    np.testing.assert_allclose(ssum[0].numpy(), ref.detach().double().sum((0, 2, 3)).numpy(), rtol=2e-3, atol=2e-3 * ref.abs().sum().item() / Co)
    np.testing.assert_allclose(ssum[1].numpy(), (ref.detach().double() ** 2).sum((0, 2, 3)).numpy(), rtol=2e-3)
    dy = bf(torch.randn(ref.shape, generator=g))
    ref.backward(dy)
    if Ci % 64 == 0 and Co % 64 == 0:
        dx = ops.conv2d_bwd_data(nhwc(dy), ops.pack_weight(w.cuda(), d, "bwd"), d)
        close(nchw(dx), xr.grad, 1.2e-2)
    dw = ops.conv2d_bwd_weight(nhwc(dy), nhwc(x), d)          # [Co][T][Ci] fp32
    close(dw.cpu().reshape(Co, K, K, Ci).permute(0, 3, 1, 2), wr.grad, 2e-3)
    dw2 = ops.conv2d_bwd_weight(nhwc(dy), nhwc(x), d, dw=dw.clone())   # accumulate path
    close(dw2.cpu(), 2 * dw.cpu(), 1e-4)


def test_conv_epilogue_res_bias_relu_f32(dev):
    from uda_poseestimation_amd import ops
    g = torch.Generator().manual_seed(2)
    N, H, W, Ci = 2, 8, 8, 256
    for Co in (16, 18, 21):      # heads: human / animal / hand key-point counts
        x = bf(torch.randn(N, Ci, H, W, generator=g))
        w = bf(torch.randn(Co, Ci, 1, 1, generator=g) * 0.05)
        b = torch.randn(Co, generator=g)
        d = ops.conv_desc(N, H, W, Ci, Co, 1)
        y = ops.conv2d_fwd(nhwc(x), ops.pack_weight(w.cuda(), d), d, bias=b.cuda(), out_f32=True)
        close(nchw(y), F.conv2d(x, w, b), 1e-4)
    Co = 64
    x = bf(torch.randn(N, Ci, H, W, generator=g))
    w = bf(torch.randn(Co, Ci, 1, 1, generator=g) * 0.05)
    r = bf(torch.randn(N, Co, H, W, generator=g))
    d = ops.conv_desc(N, H, W, Ci, Co, 1)
    y = ops.conv2d_fwd(nhwc(x), ops.pack_weight(w.cuda(), d), d, res=nhwc(r), relu=True)
    close(nchw(y), F.relu(F.conv2d(x, w) + r), 1.2e-2)


def test_stem_conv7x7(dev):
    from uda_poseestimation_amd import ops
    g = torch.Generator().manual_seed(3)
    N, H, W = 2, 64, 64
    x = bf(torch.randn(N, 3, H, W, generator=g))
    w = bf(torch.randn(64, 3, 7, 7, generator=g) * 0.1)
    d = ops.conv_desc(N, H, W, 8, 64, 7, 2, 3)
    x8 = ops.to_nhwc_bf16(x.cuda(), 8)
    assert x8.shape == (N, H, W, 8) and float(x8[..., 3:].abs().max()) == 0.0
    wr = w.clone().requires_grad_(True)
    ref = F.conv2d(x, wr, stride=2, padding=3)
    y = ops.conv2d_fwd(x8, ops.pack_weight(w.cuda(), d), d)
    close(nchw(y), ref.detach(), 1.2e-2)
    dy = bf(torch.randn(ref.shape, generator=g))
    ref.backward(dy)
    dw = ops.conv2d_bwd_weight(nhwc(dy), x8, d).cpu().reshape(64, 7, 8, 8)[:, :, :7, :3].permute(0, 3, 1, 2)
    close(dw, wr.grad, 2e-3)


@pytest.mark.parametrize("shape", [(2, 8, 8, 2048, 256), (2, 16, 16, 256, 256), (1, 5, 6, 64, 64)], ids=["up1", "up2", "odd"])
def test_deconv4x4s2(dev, shape):
    from uda_poseestimation_amd import ops
    N, H, W, Ci, Co = shape
    g = torch.Generator().manual_seed(4)
    x = bf(torch.randn(N, Ci, H, W, generator=g))
    w = bf(torch.randn(Ci, Co, 4, 4, generator=g) / (Ci * 4) ** 0.5)
    d = ops.conv_desc(N, H, W, Ci, Co, 4, 2, 1, transposed=True)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    ref = F.conv_transpose2d(xr, wr, stride=2, padding=1)
    y, stats = ops.conv2d_fwd(nhwc(x), ops.pack_weight(w.cuda(), d, "fwd"), d, want_stats=True)
    assert tuple(y.shape) == (N, 2 * H, 2 * W, Co)
    close(nchw(y), ref.detach(), 1.2e-2)
    np.testing.assert_allclose(stats.double().sum(0)[1].cpu().numpy(), (ref.detach().double() ** 2).sum((0, 2, 3)).numpy(), rtol=2e-3)
    dy = bf(torch.randn(ref.shape, generator=g))
    ref.backward(dy)
    dx = ops.conv2d_bwd_data(nhwc(dy), ops.pack_weight(w.cuda(), d, "bwd"), d)
    close(nchw(dx), xr.grad, 1.2e-2)
    dw = ops.conv2d_bwd_weight(nhwc(dy), nhwc(x), d)           # [Ci][16][Co]
    close(dw.cpu().reshape(Ci, 4, 4, Co).permute(0, 3, 1, 2), wr.grad, 2e-3)


def test_reflect_and_upsample_conv(dev):
    from uda_poseestimation_amd import ops
    g = torch.Generator().manual_seed(5)
    N, H, W, Ci, Co = 2, 10, 12, 64, 64
    x = bf(torch.randn(N, Ci, H, W, generator=g))
    w = bf(torch.randn(Co, Ci, 3, 3, generator=g) / 24)
    b = torch.randn(Co, generator=g)
    d = ops.conv_desc(N, H, W, Ci, Co, 3, 1, 1, reflect=True)
    y = ops.conv2d_fwd(nhwc(x), ops.pack_weight(w.cuda(), d), d, bias=b.cuda(), relu=True)
    close(nchw(y), F.relu(F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), w, b)), 1.2e-2)
    d2 = ops.conv_desc(N, H, W, Ci, Co, 3, 1, 1, reflect=True, upsample=True)
    y2 = ops.conv2d_fwd(nhwc(x), ops.pack_weight(w.cuda(), d2), d2, bias=b.cuda())
    xu = F.interpolate(x, scale_factor=2, mode="nearest")
    close(nchw(y2), F.conv2d(F.pad(xu, (1, 1, 1, 1), mode="reflect"), w, b), 1.2e-2)
    # 3-channel image input (first VGG 3x3 conv): Ci padded to 8
    x3 = bf(torch.randn(N, 3, H, W, generator=g))
    w3 = bf(torch.randn(64, 3, 3, 3, generator=g) / 5)
    d3 = ops.conv_desc(N, H, W, 8, 64, 3, 1, 1, reflect=True)
    y3 = ops.conv2d_fwd(ops.to_nhwc_bf16(x3.cuda(), 8), ops.pack_weight(w3.cuda(), d3), d3)
    close(nchw(y3), F.conv2d(F.pad(x3, (1, 1, 1, 1), mode="reflect"), w3), 1.2e-2)


def test_batchnorm_train_fwd_bwd(dev):
    from uda_poseestimation_amd import ops
    g = torch.Generator().manual_seed(6)
    N, H, W, Ci, C_ = 4, 16, 16, 64, 128
    x = bf(torch.randn(N, Ci, H, W, generator=g))
    w = bf(torch.randn(C_, Ci, 1, 1, generator=g) * 0.2)
    res = bf(torch.randn(N, C_, H, W, generator=g))
    gamma = torch.rand(C_, generator=g) + 0.5
    beta = torch.randn(C_, generator=g) * 0.1
    d = ops.conv_desc(N, H, W, Ci, C_, 1)
    y, stats = ops.conv2d_fwd(nhwc(x), ops.pack_weight(w.cuda(), d), d, want_stats=True)
    rm, rv = torch.zeros(C_).cuda(), torch.ones(C_).cuda()
    nbt = torch.zeros((), dtype=torch.int64).cuda()
    z, mean, invstd = ops.bn_train_fwd(y, stats, gamma.cuda(), beta.cuda(), rm, rv, nbt, res=nhwc(res), relu=True)
    # oracle on the stored bf16 conv output
    yc = nchw(y).requires_grad_(True)
    bn = torch.nn.BatchNorm2d(C_)
    with torch.no_grad():
        bn.weight.copy_(gamma); bn.bias.copy_(beta)
    bn.train()
    zr = F.relu(bn(yc) + res)
    close(nchw(z), zr.detach(), 1.5e-2)
    np.testing.assert_allclose(rm.cpu().numpy(), bn.running_mean.numpy(), rtol=2e-2, atol=2e-3)
    np.testing.assert_allclose(rv.cpu().numpy(), bn.running_var.numpy(), rtol=2e-2)
    assert int(nbt) == 1
    dz = bf(torch.randn(zr.shape, generator=g))
    # the ReLU mask is taken from the DEVICE output z (pre-activations within rounding of 0 may differ in sign
    # between the two evaluation orders; the gradient definition is the same)
    gm = dz * (nchw(z) > 0)
    (bn(yc) + res).backward(gm)
    dy, dgamma, dbeta, gmask = ops.bn_bwd(nhwc(dz), z, y, gamma.cuda(), mean, invstd, relu=True, want_g=True)
    close(nchw(dy), yc.grad, 1.2e-2)
    close(dgamma.cpu(), bn.weight.grad, 5e-3)
    close(dbeta.cpu(), bn.bias.grad, 5e-3)
    close(nchw(gmask), gm, 1.2e-2)
    assert ((nchw(z) > 0) != (zr.detach() > 0)).float().mean().item() < 1e-3
    # BN + ReLU without a residual: the mask recomputed from y (relu=2, z not read) is the mask of the stored output
    z2, mean2, invstd2 = ops.bn_train_fwd(y, stats, gamma.cuda(), beta.cuda(), rm, rv, nbt, res=None, relu=True)
    a = ops.bn_bwd(nhwc(dz), z2, y, gamma.cuda(), mean2, invstd2, relu=1, want_g=True)
    b = ops.bn_bwd(nhwc(dz), None, y, gamma.cuda(), mean2, invstd2, relu=2, want_g=True, beta=beta.cuda())
    for u, v in zip(a, b):
        assert torch.equal(u, v)


@pytest.mark.parametrize("shape", [(4, 32, 256, True), (8, 16, 512, False), (3, 20, 1024, False), (2, 24, 2048, True)])
def test_batchnorm_backward_channel_chunked(dev, shape):
    """The channel-chunked BN backward (C >= 256, 1 K..32 K pixels: reduce + apply with the finalize folded into every apply
    work-group's prelude) against the closed form in fp64 on identical tensors; bf16 and fp32 incoming gradients, all three
    ReLU-mask modes, ragged pixel counts (1200, 1152 pixels: partial last range)."""
    from uda_poseestimation_amd import ops
    N, H, C_, f32 = shape
    g = torch.Generator().manual_seed(10 + C_)
    y = torch.randn(N, H, H, C_, generator=g).bfloat16().cuda()
    gamma = (torch.rand(C_, generator=g) + 0.5).cuda()
    beta = (torch.randn(C_, generator=g) * 0.1).cuda()
    yf = y.double()
    mean = yf.mean((0, 1, 2)); var = yf.var((0, 1, 2), unbiased=False)
    invstd = (1.0 / torch.sqrt(var + 1e-5))
    mean_f, invstd_f = mean.float(), invstd.float()
    sc = gamma * invstd_f; sh = beta - mean_f * sc
    z = torch.relu(y.float() * sc + sh).bfloat16()
    dz = torch.randn(N, H, H, C_, generator=g).cuda()
    dzk = dz if f32 else dz.bfloat16()
    M = N * H * H
    for mode in (0, 1, 2):
        gm = dzk.double() * ((z.float() > 0) if mode else torch.ones_like(z, dtype=torch.bool))
        xh = (yf - mean_f.double()) * invstd_f.double()
        dbeta = gm.sum((0, 1, 2)); dgamma = (gm * xh).sum((0, 1, 2))
        dy = (gamma * invstd_f).double() * (gm - dbeta / M - xh * dgamma / M)
        o = ops.bn_bwd(dzk, z if mode == 1 else None, y, gamma, mean_f, invstd_f, relu=mode, want_g=True, beta=beta)
        scale = float(dy.abs().max())
        assert float((o[0].double() - dy).abs().max()) <= 6e-3 * scale + 1e-6          # bf16 output rounding
        assert float((o[1].double() - dgamma).abs().max()) <= 2e-5 * float(dgamma.abs().max()) + 1e-4
        assert float((o[2].double() - dbeta).abs().max()) <= 2e-5 * float(dbeta.abs().max()) + 1e-4
        assert float((o[3].double() - gm).abs().max()) <= 4e-3 * float(gm.abs().max())


def test_maxpool(dev):
    from uda_poseestimation_amd import ops
    g = torch.Generator().manual_seed(7)
    x = F.relu(bf(torch.randn(2, 64, 18, 20, generator=g)))        # many exact ties at 0 (post-ReLU)
    xr = x.clone().requires_grad_(True)
    ref = F.max_pool2d(xr, 3, 2, 1)
    y, idx = ops.maxpool3x3s2_fwd(nhwc(x))
    assert torch.equal(nchw(y), ref.detach())
    dy = bf(torch.randn(ref.shape, generator=g))
    ref.backward(dy)
    dx = ops.maxpool3x3s2_bwd(nhwc(dy), idx, 18, 20)
    close(nchw(dx), xr.grad, 1.2e-2)
    x2 = bf(torch.randn(2, 64, 9, 11, generator=g))
    assert torch.equal(nchw(ops.maxpool2x2_ceil(nhwc(x2))), F.max_pool2d(x2, 2, 2, 0, ceil_mode=True))


def test_adain_matches_oracle(dev, golden_dir):
    import os
    from oracle import style_ref
    from uda_poseestimation_amd import ops
    z = np.load(os.path.join(golden_dir, "style.npz"))
    c, s = bf(torch.from_numpy(z["c"])), bf(torch.from_numpy(z["s"]))
    out, st = ops.adain(nhwc(c), nhwc(s), alpha=0.6, want_stats=True)
    ref = 0.6 * style_ref.adain_ref(c, s) + 0.4 * c
    close(nchw(out), ref, 1.2e-2)
    m, sd = style_ref.calc_mean_std_ref(c)
    np.testing.assert_allclose(st[..., 0].cpu().numpy(), m.reshape(2, 512).numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(st[..., 1].cpu().numpy(), sd.reshape(2, 512).numpy(), rtol=1e-4)


@pytest.mark.parametrize("case", [("1x1", 2, 16, 16, 64, 256, 1, 1, 0), ("3x3_s2", 2, 16, 16, 128, 128, 3, 2, 1), ("3x3_ragged", 3, 12, 12, 32, 64, 3, 1, 1)],
                         ids=lambda c: c[0])
def test_conv_fp32_exact_path(dev, case):
    """Exact fp32 MFMA path (forward only): fp32 in / fp32 out, compared with torch fp32 at 2e-6 relative."""
    from uda_poseestimation_amd import ops
    _, N, H, W, Ci, Co, K, s, p = case
    g = torch.Generator().manual_seed(11)
    x = torch.randn(N, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, K, K, generator=g) / (Ci * K * K) ** 0.5
    r = torch.randn(N, Co, (H + 2 * p - K) // s + 1, (W + 2 * p - K) // s + 1, generator=g)
    d = ops.conv_desc(N, H, W, Ci, Co, K, s, p)
    xg = x.permute(0, 2, 3, 1).contiguous().cuda()
    wg = w.permute(0, 2, 3, 1).contiguous().reshape(Co, K * K, Ci).cuda()
    rg = r.permute(0, 2, 3, 1).contiguous().cuda()
    y, stats = ops.conv2d_fwd(xg, wg, d, res=rg, relu=True, want_stats=True)
    assert y.dtype == torch.float32
    ref = F.conv2d(x, w, stride=s, padding=p)
    close(nchw(y), F.relu(ref + r), 2e-6)
    np.testing.assert_allclose(stats.double().sum(0)[0].cpu().numpy(), ref.double().sum((0, 2, 3)).numpy(), rtol=1e-4, atol=1e-3)


FULL_SIZE_LAYERS = [
    # name, H, Ci, Co, K, stride, pad, transposed      (N = 32: the shapes bench.py runs; too large for the CPU oracle)
    ("l1.c3 64->256 1x1 @64", 64, 64, 256, 1, 1, 0, False),
    ("l3.c2 256->256 3x3 @16", 16, 256, 256, 3, 1, 1, False),
    ("l3.c1 1024->256 1x1 @16", 16, 1024, 256, 1, 1, 0, False),
    ("l2.c2 128->128 3x3 s2 @64", 64, 128, 128, 3, 2, 1, False),
    ("up2 deconv 256->256 4x4 s2 @32", 32, 256, 256, 4, 2, 1, True),
]


@pytest.mark.parametrize("layer", FULL_SIZE_LAYERS, ids=[l[0] for l in FULL_SIZE_LAYERS])
def test_conv_adjoint_identities_at_full_size(dev, layer):
    """Size-independent property at BASELINE.json's sizes (N = 32): forward, data-gradient and weight-gradient kernels are
    the three faces of one bilinear form, so <conv(x, w), dy> = <x, dgrad(dy, w)> = <w, wgrad(dy, x)> up to bf16 output
    rounding.  Also linearity: conv(x1 + x2) = conv(x1) + conv(x2) within rounding."""
    from uda_poseestimation_amd import ops
    _, H, Ci, Co, K, s, p, tr = layer
    N = 32
    g = torch.Generator(device="cuda").manual_seed(3)
    d = ops.conv_desc(N, H, H, Ci, Co, K, s, p, transposed=tr)
    ho, wo = ops.conv_out_hw(d)
    x = torch.randn(N, H, H, Ci, device="cuda", generator=g).bfloat16()
    wshape = (Ci, Co, K, K) if tr else (Co, Ci, K, K)
    w = (torch.randn(wshape, device="cuda", generator=g) / (Ci * K * K) ** 0.5).bfloat16().float()
    dy = torch.randn(N, ho, wo, Co, device="cuda", generator=g).bfloat16()
    wf, wb = ops.pack_weight(w, d, "fwd"), ops.pack_weight(w, d, "bwd")
    y = ops.conv2d_fwd(x, wf, d, out_f32=True)
    dx = ops.conv2d_bwd_data(dy, wb, d, out_f32=True)
    dw = ops.conv2d_bwd_weight(dy, x, d)                       # fp32 [Co][T][Ci] (transposed: [Ci][T][Co])
    a = float((y.double() * dy.double()).sum())
    b = float((dx.double() * x.double()).sum())
    w_phys = (w.permute(0, 2, 3, 1).reshape(dw.shape)).double()   # [A][KH*KW][B], the layout dw uses
    c = float((dw.double() * w_phys).sum())
    scale = float(y.double().norm() * dy.double().norm())
    assert abs(a - b) <= 1e-4 * scale and abs(a - c) <= 1e-4 * scale, (a, b, c, scale)
    x2 = torch.randn(N, H, H, Ci, device="cuda", generator=g).bfloat16()
    xs = (x.float() + x2.float()).bfloat16()
    lhs = ops.conv2d_fwd(xs, wf, d, out_f32=True)
    rhs = y + ops.conv2d_fwd(x2, wf, d, out_f32=True)
    # (x + x2 is rounded to bf16 once more than x and x2: 2^-9 relative per input element)
    assert float((lhs - rhs).norm() / rhs.norm()) <= 6e-3


DGRAD_BN_CASES = [
    # name, N, H, Ci, Co, K, stride, pad, transposed, mask-from-z, residual, fp32 out
    ("1x1_c3_mask_y", 4, 16, 64, 256, 1, 1, 0, False, False, False, False),          # dgrad c3 -> bn2
    ("3x3_c2_mask_y", 3, 12, 64, 64, 3, 1, 1, False, False, False, False),           # dgrad c2 -> bn1 (ragged M)
    ("3x3_s2_subpixel_odd", 2, 15, 128, 128, 3, 2, 1, False, False, False, False),   # stride-2 dgrad: 4 classes, odd size
    ("1x1_c1_skip_mask_z", 4, 16, 256, 64, 1, 1, 0, False, True, True, False),       # dgrad c1 + skip -> bn3 of the block before
    ("deconv_f32_mask_y", 2, 8, 256, 256, 4, 2, 1, True, False, False, True),        # dgrad of a deconv -> deconv BN (fp32 g)
    ("head_like_f32", 2, 32, 256, 64, 1, 1, 0, False, False, False, True),           # head dgrad (64 padded channels) -> fp32
    ("n32_l3_c1_mask_z", 32, 16, 1024, 256, 1, 1, 0, False, True, True, False),      # a bench-size launch (128x64 tiles, 64 rows)
]


@pytest.mark.parametrize("case", DGRAD_BN_CASES, ids=[c[0] for c in DGRAD_BN_CASES])
def test_dgrad_with_fused_bn_backward_reduction(dev, case):
    """udapose_conv2d_bwd_data_bn against udapose_conv2d_bwd_data: the masked output equals the plain dgrad output under the
    consumer BatchNorm's ReLU mask BIT FOR BIT, and the slab's column sums equal sum(g), sum(g * xhat) of exactly those stored
    values (fp64 reference; fp32 partials per m-tile)."""
    from uda_poseestimation_amd import ops
    _, N, H, Ci, Co, K, s, p, tr, mask_z, with_res, f32 = case
    g = torch.Generator(device="cuda").manual_seed(11)
    d = ops.conv_desc(N, H, H, Ci, Co, K, s, p, transposed=tr)
    ho, wo = ops.conv_out_hw(d)
    wshape = (Ci, Co, K, K) if tr else (Co, Ci, K, K)
    w = (torch.randn(wshape, device="cuda", generator=g) / (Co * K * K) ** 0.5).bfloat16().float()
    wb = ops.pack_weight(w, d, "bwd")
    dy = torch.randn(N, ho, wo, Co, device="cuda", generator=g).bfloat16()
    res = torch.randn(N, H, H, Ci, device="cuda", generator=g).bfloat16() if with_res else None
    bn_y = (torch.randn(N, H, H, Ci, device="cuda", generator=g) * 1.5 + 0.3).bfloat16()
    mean = torch.randn(Ci, device="cuda", generator=g) * 0.2 + 0.3
    invstd = torch.rand(Ci, device="cuda", generator=g) * 0.5 + 0.4
    gamma = torch.rand(Ci, device="cuda", generator=g) + 0.5
    beta = torch.randn(Ci, device="cuda", generator=g) * 0.3
    bn_z = torch.relu(torch.randn(N, H, H, Ci, device="cuda", generator=g)).bfloat16() if mask_z else None
    dx0 = ops.conv2d_bwd_data(dy, wb, d, res=res, out_f32=f32)
    gq, slab = ops.conv2d_bwd_data_bn(dy, wb, d, bn_y, mean, invstd, bn_z=bn_z, bn_gamma=gamma, bn_beta=beta, res=res, out_f32=f32)
    assert gq.dtype == dx0.dtype and slab.shape[1:] == (2, Ci)
    if mask_z:
        keep = bn_z.float() > 0
        sure = torch.ones_like(keep)
    else:
        sc = gamma * invstd
        t = bn_y.double() * sc.double() + (beta - mean * sc).double()
        keep = t > 0
        sure = t.abs() > 1e-5 * (bn_y.double().abs() * sc.double().abs() + 1.0)      # (fma vs mul+add can differ only at t ~ 0)
    want = torch.where(keep, dx0, torch.zeros_like(dx0))
    assert sure.float().mean() > 0.999
    assert torch.equal(gq[sure], want[sure]), "masked dgrad output differs from mask * plain dgrad output"
    assert 0.2 < float((gq != 0).float().mean()) < 0.8
    xhat = (bn_y.double() - mean.double()) * invstd.double()
    s1 = gq.double().sum((0, 1, 2))
    s2 = (gq.double() * xhat).sum((0, 1, 2))
    a1 = gq.double().abs().sum((0, 1, 2))
    a2 = (gq.double() * xhat).abs().sum((0, 1, 2))
    got = slab.double().sum(0)
    assert float(((got[0] - s1).abs() / (a1 + 1e-30)).max()) < 2e-6, float(((got[0] - s1).abs() / (a1 + 1e-30)).max())
    assert float(((got[1] - s2).abs() / (a2 + 1e-30)).max()) < 2e-6, float(((got[1] - s2).abs() / (a2 + 1e-30)).max())
    # the BatchNorm backward that consumes (g, slab) without a reduction pass (udapose_bn_bwd_pre): closed form in fp64 on the
    # same stored g and y; dgamma / dbeta are the column sums themselves
    dyq, dgamma, dbeta = ops.bn_bwd_pre(gq, bn_y, gamma, mean, invstd, slab)
    npix = N * H * H
    ref = (gamma * invstd).double() * (gq.double() - s1 / npix - xhat * (s2 / npix))
    close(dyq.double().cpu(), ref.cpu(), 6e-3)
    np.testing.assert_allclose(dbeta.double().cpu().numpy(), s1.cpu().numpy(), rtol=0, atol=2e-6 * float(a1.max()))
    np.testing.assert_allclose(dgamma.double().cpu().numpy(), s2.cpu().numpy(), rtol=0, atol=2e-6 * float(a2.max()))


H3_CASES = [
    # name, N, H, Ci, Co      (3x3, stride 1, pad 1)
    ("w16", 2, 16, 64, 64), ("w12_ragged", 3, 12, 64, 128), ("w8_two_chunks", 2, 8, 128, 64), ("w32", 1, 32, 64, 64),
    ("w64", 1, 64, 64, 64), ("w10_m300", 3, 10, 64, 64), ("w16_four_chunks", 1, 16, 256, 128), ("n32_l3", 32, 16, 256, 256),
]


@pytest.mark.parametrize("mode", [2, 3], ids=["rows64", "rows128"])
@pytest.mark.parametrize("case", H3_CASES, ids=[c[0] for c in H3_CASES])
def test_run_staged_3x3_form(dev, case, mode):
    """The run-staged form of 3x3 stride-1 pad-1 convolutions (igemm.hip, H3: the A operand staged once per 64 channels as a
    run of consecutive pixels, taps read at row offsets, padded taps from a zero row) in both tile heights, fprop with the
    BatchNorm statistics and the data gradient (plain, and with the consumer BatchNorm's mask + sums), against torch on the
    same bf16-rounded inputs and against the tap-staged form (mode 0): same products, other summation order."""
    from uda_poseestimation_amd import ops, _hip
    lib = _hip.lib()
    _, N, H, Ci, Co = case
    g = torch.Generator().manual_seed(4)
    big = N * H * H * Ci > 1 << 21
    x = bf(torch.randn(N, Ci, H, H, generator=g))
    w = bf(torch.randn(Co, Ci, 3, 3, generator=g) / (Ci * 9) ** 0.5)
    dy = bf(torch.randn(N, Co, H, H, generator=g))
    d = ops.conv_desc(N, H, H, Ci, Co, 3, 1, 1)
    wf, wb = ops.pack_weight(w.cuda(), d, "fwd"), ops.pack_weight(w.cuda(), d, "bwd")
    bn_y = (torch.randn(N, H, H, Ci, generator=g) + 0.2).bfloat16().cuda()
    mean, invstd, gamma, beta = (torch.rand(Ci, generator=g).cuda() + 0.5 for _ in range(4))
    out = {}
    for m in (0, mode):
        dm = ops.with_policy(d, _hip.policy(igemm_h3=m))       # the form is chosen by the call's explicit policy
        y, stats = ops.conv2d_fwd(nhwc(x), wf, dm, want_stats=True)
        dx = ops.conv2d_bwd_data(nhwc(dy), wb, dm)
        gq, slab = ops.conv2d_bwd_data_bn(nhwc(dy), wb, dm, bn_y, mean, invstd, bn_gamma=gamma, bn_beta=beta)
        out[m] = (y.float(), stats.double().sum(0), dx.float(), gq.float(), slab.double().sum(0))
    for a, b in zip(out[0], out[mode]):
        scale = float(a.abs().max()) + 1e-12
        assert float((a - b).abs().max()) <= 8e-3 * scale, (float((a - b).abs().max()), scale)
    if not big:
        xr = x.clone().requires_grad_(True)
        ref = F.conv2d(xr, w, padding=1)
        ref.backward(dy)
        close(nchw(out[mode][0].cuda()), ref.detach(), 1.2e-2)
        close(nchw(out[mode][2].cuda()), xr.grad, 1.2e-2)
        np.testing.assert_allclose(out[mode][1][0].cpu().numpy(), ref.detach().double().sum((0, 2, 3)).numpy(), rtol=2e-3,
                                   atol=2e-3 * ref.abs().sum().item() / Co)


@pytest.mark.parametrize("case", [(4, 16, 16, 256, 256), (2, 64, 64, 64, 64), (8, 8, 8, 512, 128), (3, 32, 32, 128, 64), (2, 16, 8, 64, 192)])
def test_weight_gradient_filter_row_form_3x3(case):
    """Policy wgrad_row3: the weight gradient of a 3x3 stride-1 pad-1 convolution with one work-group per (64x64 tile, filter row) -
    the row's three taps share one staged dy tile and one x window, the zero padding is a zero dy row (rows) and lane masks on the
    dy fragments (column wrap).  Against torch's fp32 weight gradient on the same bf16-rounded operands, and against the one-tap
    form; split reductions and the accumulate mode included."""
    from uda_poseestimation_amd import ops, _hip
    N, H, W, Ci, Co = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, Ci, H, W, generator=g).bfloat16().float()
    dy = torch.randn(N, Co, H, W, generator=g).bfloat16().float()
    xr = x.clone().requires_grad_(False)
    w = torch.zeros(Co, Ci, 3, 3, requires_grad=True)
    torch.nn.functional.conv2d(xr, w, padding=1).backward(dy)
    ref = w.grad.permute(0, 2, 3, 1).reshape(Co, 9, Ci)                       # [Co][tap][Ci]
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().bfloat16().cuda()
    d = ops.conv_desc(N, H, W, Ci, Co, 3, 1, 1)
    scale = ref.abs().max().item()
    outs = {}
    for row3 in (1, 0):
        for ks in (0, 4):
            dd = ops.with_policy(d, _hip.policy(wgrad_row3=row3, wgrad_ksplit=ks if ks else -1))
            dw = ops.conv2d_bwd_weight(nhwc(dy), nhwc(x), dd)
            err = (dw.cpu() - ref).abs().max().item()
            assert err <= 2e-3 * scale, (row3, ks, err, scale)
            outs[(row3, ks)] = dw
    assert (outs[(1, 0)] - outs[(0, 0)]).abs().max().item() <= 1e-3 * scale
    # every tap, border columns included: the wrap masks (dx = -1 at j = 0, dx = +1 at j = W - 1) change the result where they act
    for t in range(9):
        assert (outs[(1, 0)][:, t].cpu() - ref[:, t]).abs().max().item() <= 2e-3 * scale, t
    dd = ops.with_policy(d, _hip.policy(wgrad_row3=1))
    acc = ops.conv2d_bwd_weight(nhwc(dy), nhwc(x), dd, dw=outs[(1, 0)].clone())
    assert (acc.cpu() - 2 * ref).abs().max().item() <= 4e-3 * scale


@pytest.mark.parametrize("case", [(4, 16, 16, 256, 256, 3), (2, 64, 64, 64, 64, 3), (8, 8, 8, 512, 128, 3), (3, 32, 32, 128, 64, 3),
                                  (4, 16, 16, 1024, 256, 1), (2, 32, 32, 128, 512, 1), (2, 16, 8, 64, 192, 3)])
def test_weight_gradient_buffer_load_loader_is_bit_identical(case):
    """Policy wgrad_fastgeo = 2: the fast-geometry weight-gradient loop with buffer_load ... lds (constant lane offsets, scalar
    stage offsets, out-of-range offsets for taps outside the image: the hardware writes the zeros) and the ring unrolled over its
    buffers.  Same tiles, same MFMA order: bit-identical to the pointer-select loader (one pixel split), and within fp32 atomic
    reordering of it when the pixel reduction is split; both forms against torch."""
    from uda_poseestimation_amd import ops, _hip
    N, H, W, Ci, Co, K = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, Ci, H, W, generator=g).bfloat16().float()
    dy = torch.randn(N, Co, H, W, generator=g).bfloat16().float()
    w = torch.zeros(Co, Ci, K, K, requires_grad=True)
    torch.nn.functional.conv2d(x, w, padding=K // 2).backward(dy)
    ref = w.grad.permute(0, 2, 3, 1).reshape(Co, K * K, Ci)
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().bfloat16().cuda()
    d = ops.conv_desc(N, H, W, Ci, Co, K, 1, K // 2)
    scale = ref.abs().max().item()
    out = {}
    for fg in (1, 2):
        for tile in (0, 1):
            if tile == 0 and (Ci < 128 or Co < 128):
                continue
            for ks in (1, 4):
                dd = ops.with_policy(d, _hip.policy(wgrad_fastgeo=fg, wgrad_tile=tile, wgrad_ksplit=ks, wgrad_row3=0))
                dw = ops.conv2d_bwd_weight(nhwc(dy), nhwc(x), dd)
                assert (dw.cpu() - ref).abs().max().item() <= 2e-3 * scale, (fg, tile, ks)
                out[(fg, tile, ks)] = dw
    for (fg, tile, ks), dw in out.items():
        if fg == 2:
            other = out[(1, tile, ks)]
            if ks == 1:
                assert torch.equal(dw, other), (tile, ks)
            else:
                assert (dw - other).abs().max().item() <= 1e-4 * scale
