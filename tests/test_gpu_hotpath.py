"""GPU parity of the heat-map / mean-teacher pieces of the hot path, through the reference's own API names, against the
golden vectors captured from the reference (tests/golden) and against the CPU oracle.  Integer / index results are
bit-exact; fp32 reductions are compared at 1e-6 relative (different summation order)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    """A TCP port nobody listens on right now (the two-rank tests must not collide when test modules run side by side)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


@pytest.fixture(autouse=True)
def _bf16_unless_stated(monkeypatch):
    """The tests of this module exercise the 16-bit executor: networks start in 'bf16' (BASELINE.json's benched precision) unless a
    test sets another precision.  (A new module's default is 'auto': autocast dtype / fp32-grade teacher, tests/test_gpu_dropin_loop.py.)"""
    from uda_poseestimation_amd.lib.models.pose_resnet import PoseResNet
    monkeypatch.setattr(PoseResNet, "default_precision", "bf16")

def _g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_losses_match_reference_goldens_and_gradients(golden_dir):
    from uda_poseestimation_amd.lib.models.loss import ConsLoss, JointsMSELoss
    from oracle import losses_ref
    z = _g(golden_dir, "losses.npz")
    pred, gt, w = (torch.from_numpy(z[k]).cuda() for k in ("pred", "gt", "w"))
    mask = torch.from_numpy(z["mask"]).cuda()
    np.testing.assert_allclose(JointsMSELoss()(pred, gt, w).item(), z["mse_mean"], rtol=1e-6)
    np.testing.assert_allclose(JointsMSELoss()(pred, gt).item(), z["mse_mean_now"], rtol=1e-6)
    np.testing.assert_allclose(JointsMSELoss(reduction="none")(pred, gt, w).cpu().numpy(), z["mse_none"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(ConsLoss()(pred, gt, tea_mask=mask).item(), z["cons_masked"], rtol=1e-6)
    np.testing.assert_allclose(ConsLoss()(pred, gt).item(), z["cons_plain"], rtol=1e-6)
    assert JointsMSELoss(reduction="sum")(pred, gt, w) is None          # the reference's silent fall-through
    with pytest.raises(IndexError):                                      # a [B,K] mask cannot index loss_map [B,H,W] (torch raises the same)
        ConsLoss()(pred, gt, valid_mask=mask)
    # gradients vs autograd of the oracle
    p1 = pred.clone().requires_grad_(True)
    (3.0 * JointsMSELoss()(p1, gt, w)).backward()
    p2 = torch.from_numpy(z["pred"]).requires_grad_(True)
    (3.0 * losses_ref.joints_mse_ref(p2, torch.from_numpy(z["gt"]), torch.from_numpy(z["w"]))).backward()
    np.testing.assert_allclose(p1.grad.cpu().numpy(), p2.grad.numpy(), rtol=1e-5, atol=1e-10)
    s1 = pred.clone().requires_grad_(True)
    (0.5 * ConsLoss()(s1, gt, tea_mask=mask)).backward()
    s2 = torch.from_numpy(z["pred"]).requires_grad_(True)
    (0.5 * losses_ref.cons_loss_ref(s2, torch.from_numpy(z["gt"]), tea_mask=torch.from_numpy(z["mask"]))).backward()
    np.testing.assert_allclose(s1.grad.cpu().numpy(), s2.grad.numpy(), rtol=1e-5, atol=1e-10)


@pytest.mark.parametrize("B,K,H,W", [(1, 21, 56, 56), (3, 17, 24, 40), (5, 1, 7, 9), (2, 33, 96, 72)], ids=["b1_k21_56", "b3_k17_24x40", "b5_k1_7x9", "b2_k33_96x72"])
def test_heatmap_kernels_on_ragged_shapes_match_the_oracle(B, K, H, W):
    """Losses (values and gradients), decode, rectify, activations + rectify in one sweep, the confidence mask and PCK on shapes off the
    benchmark's grid: one image, one key point, key-point counts that are not multiples of 8, maps whose pixel count is not a multiple of
    the 256-thread sweeps (63 pixels) and non-square maps - against the CPU oracle (integer / index results bit-exact)."""
    from oracle import losses_ref
    from oracle.keypoints_ref import accuracy_ref, get_max_preds_ref, get_max_preds_torch_ref
    from oracle.mean_teacher_ref import conf_mask_ref, rectify_ref
    from uda_poseestimation_amd import utils as U
    from uda_poseestimation_amd.lib import keypoint_detection as kd
    from uda_poseestimation_amd.lib.models.loss import ConsLoss, JointsMSELoss
    g = torch.Generator().manual_seed(B * 1000 + K)
    pred, gt = torch.rand(B, K, H, W, generator=g), torch.rand(B, K, H, W, generator=g)
    w = (torch.rand(B, K, 1, generator=g) > 0.3).float()
    mask = torch.rand(B, K, generator=g) > 0.4
    for dev_fn, ref_fn in ((lambda p_: JointsMSELoss()(p_, gt.cuda(), w.cuda()), lambda p_: losses_ref.joints_mse_ref(p_, gt, w)),
                           (lambda p_: ConsLoss()(p_, gt.cuda(), tea_mask=mask.cuda()), lambda p_: losses_ref.cons_loss_ref(p_, gt, tea_mask=mask))):
        p1, p2 = pred.clone().cuda().requires_grad_(True), pred.clone().requires_grad_(True)
        l1, l2 = dev_fn(p1), ref_fn(p2)
        np.testing.assert_allclose(l1.item(), l2.item(), rtol=2e-6)
        l1.backward(); l2.backward()
        np.testing.assert_allclose(p1.grad.cpu().numpy(), p2.grad.numpy(), rtol=1e-5, atol=1e-10)
    hm = pred.numpy()
    p, v = kd.get_max_preds(hm)
    pr, vr = get_max_preds_ref(hm)
    np.testing.assert_array_equal(p, pr); np.testing.assert_array_equal(v, vr)
    pt, vt = U.get_max_preds_torch(pred.cuda())
    ptr_, vtr = get_max_preds_torch_ref(pred)
    np.testing.assert_array_equal(pt.cpu().numpy(), ptr_.numpy()); np.testing.assert_array_equal(vt.cpu().numpy(), vtr.numpy())
    if H == W:          # (rectify's bounds test compares x with H and y with W, utils.py:89: the oracle keeps the quirk; square maps are what the loop feeds it)
        for sigma in (2, 1.0):
            want = rectify_ref(pred, sigma).numpy()
            np.testing.assert_array_equal(U.rectify(pred.cuda(), sigma).cpu().numpy(), want)
            act, rect = U.activations_and_rectify(pred.cuda(), sigma)
            np.testing.assert_array_equal(rect.cpu().numpy(), want)
            np.testing.assert_array_equal(act.cpu().numpy(), pred.amax(dim=(2, 3)).numpy())
    if B * K >= 4:
        for ratio in (0.5, 0.25):
            m_ref, act_ref, thr_ref = conf_mask_ref(pred, ratio)
            m, act, thr = U.confidence_mask(pred.cuda(), ratio)
            assert m.dtype == torch.bool and float(thr) == thr_ref
            np.testing.assert_array_equal(m.cpu().numpy(), m_ref.numpy())
    acc, avg, cnt, _ = kd.accuracy(hm, gt.numpy())
    acc_r, avg_r, cnt_r, _ = accuracy_ref(hm, gt.numpy())
    np.testing.assert_allclose(acc, acc_r, atol=1e-6)
    assert abs(avg - avg_r) < 1e-6 and cnt == cnt_r


from helpers.warp_ties import tie_exposed as _tie_exposed


@pytest.mark.parametrize("B,C,H,W", [(1, 21, 56, 56), (3, 17, 24, 40), (5, 1, 7, 9), (2, 33, 96, 72)], ids=["b1_c21_56", "b3_c17_24x40", "b5_c1_7x9", "b2_c33_96x72"])
def test_warp_chain_on_ragged_shapes_matches_the_oracle_and_its_backward_is_the_adjoint(B, C, H, W):
    """The re-warp (three chained nearest resamplings, one launch) on channel counts 1 / 17 / 21 / 33, one image, odd and non-square maps:
    forward against the per-sample torchvision restatement (identical up to isolated nearest-neighbour ties), and the size-independent
    property of the backward: <warp(x), r> == <x, warp^T(r)> for random x, r (the scatter is the exact transpose of the gather)."""
    from oracle.affine_ref import warp3_ref
    from uda_poseestimation_amd import synthetic, warp
    g = torch.Generator().manual_seed(B * 100 + C)
    x = torch.randn(B, C, H, W, generator=g)
    ap = synthetic.aug_params(B, np.random.RandomState(C))
    angle, (tx, ty), (sx, sy), sc = ap
    y = warp.recon_heatmaps(x.cuda(), ap, 4.0).cpu()
    ref = torch.stack([warp3_ref(x[i], float(angle[i]), float(tx[i]), float(ty[i]), float(sx[i]), float(sy[i]), float(sc[i]), 4.0) for i in range(B)])
    assert y.shape == ref.shape
    assert (y != ref).float().mean().item() < 5e-3
    # ... and every pixel that differs is PROVABLY a rounding tie: its fp64 source coordinate lies within 1e-4 of a half-integer at some stage
    mism_px, exposed = (y != ref).any(dim=1), _tie_exposed(ap, B, H, W, 4.0)
    assert not bool((mism_px & ~exposed).any()), f"{int((mism_px & ~exposed).sum())} pixels differ from the oracle away from any rounding tie"
    th = warp.recon_thetas(ap, B, 4.0, "cuda")
    xd = x.cuda().requires_grad_(True)
    r = torch.randn(B, C, H, W, generator=g).cuda()
    out = warp.warp_chain(xd, th)
    lhs = (out.detach().double() * r.double()).sum().item()
    (out * r).sum().backward()
    rhs = (x.cuda().double() * xd.grad.double()).sum().item()
    assert abs(lhs - rhs) <= 1e-5 * (out.detach().abs().double() * r.abs().double()).sum().item() + 1e-9, (lhs, rhs)


def test_decode_rectify_pck_bit_exact(golden_dir):
    from uda_poseestimation_amd import utils as U
    from uda_poseestimation_amd.lib import keypoint_detection as kd
    z = _g(golden_dir, "decode.npz")
    noisy = z["noisy"]
    p, v = kd.get_max_preds(noisy)                       # numpy in -> numpy out (reference call style)
    np.testing.assert_array_equal(p, z["preds_np"])
    np.testing.assert_array_equal(v, z["maxv_np"])
    pt, vt = U.get_max_preds_torch(torch.from_numpy(noisy).cuda())
    np.testing.assert_array_equal(pt.cpu().numpy(), z["preds_t"])
    np.testing.assert_array_equal(vt.cpu().numpy(), z["maxv_t"])
    acc, avg, cnt, pk = kd.accuracy(noisy, z["labels"])
    np.testing.assert_allclose(acc, z["acc"], atol=1e-6)
    assert abs(avg - float(z["avg_acc"])) < 1e-6 and cnt == int(z["cnt"])
    np.testing.assert_array_equal(pk, z["pred_kp"])
    acc2, avg2, cnt2, _ = kd.accuracy(torch.from_numpy(noisy).cuda(), torch.from_numpy(z["labels"]).cuda())
    assert avg2 == avg and cnt2 == cnt
    np.testing.assert_array_equal(U.rectify(torch.from_numpy(noisy).cuda(), 2).cpu().numpy(), z["rect_s2"])
    np.testing.assert_array_equal(U.rectify(torch.from_numpy(noisy).cuda(), 1.0).cpu().numpy(), z["rect_s1"])
    with pytest.raises(NotImplementedError):
        U.rectify(torch.from_numpy(noisy).cuda(), 1.5)
    # larger, non-square maps (animal config: 96x96) against the CPU oracle
    from oracle.mean_teacher_ref import rectify_ref
    hm = torch.randn(3, 18, 96, 96, generator=torch.Generator().manual_seed(4))
    np.testing.assert_array_equal(U.rectify(hm.cuda(), 1.0).cpu().numpy(), rectify_ref(hm, 1.0).numpy())


def test_ema_bit_exact_with_reference(golden_dir):
    from uda_poseestimation_amd.utils import OldWeightEMA
    z = _g(golden_dir, "ema.npz")

    class Holder(torch.nn.Module):
        def __init__(self, arrs):
            super().__init__()
            self.ps = torch.nn.ParameterList([torch.nn.Parameter(torch.from_numpy(a).clone()) for a in arrs])

    stu = Holder([z[f"src{i}"] for i in range(3)]).cuda()
    tea = Holder([np.zeros_like(z[f"src{i}"]) for i in range(3)]).cuda()
    ema = OldWeightEMA(tea, stu, alpha=0.999)
    for i, p in enumerate(tea.parameters()):
        np.testing.assert_array_equal(p.detach().cpu().numpy(), z[f"init{i}"])
    for it in range(3):
        with torch.no_grad():
            for i, p in enumerate(stu.parameters()):
                p.copy_(torch.from_numpy(z[f"stu_it{it}_{i}"]))
        ema.step()
    for i, p in enumerate(tea.parameters()):
        np.testing.assert_array_equal(p.detach().cpu().numpy(), z[f"final{i}"])     # bit-exact two-rounding form


@pytest.mark.parametrize("kind", ["adam", "sgd"])
def test_fused_optimizers_match_torch(kind):
    from uda_poseestimation_amd.optim import FusedAdam, FusedSGD
    g = torch.Generator().manual_seed(0)
    shapes = [(64, 3, 7, 7), (256,), (128, 64, 3, 3), (5000,), (16, 256, 1, 1)]
    ref_p = [torch.nn.Parameter(torch.randn(s, generator=g)) for s in shapes]
    dev_p = [torch.nn.Parameter(p.detach().clone().cuda()) for p in ref_p]
    if kind == "adam":
        o_ref, o_dev = torch.optim.Adam(ref_p, lr=1e-3), FusedAdam(dev_p, lr=1e-3)
    else:
        o_ref = torch.optim.SGD(ref_p, lr=1e-2, momentum=0.9, weight_decay=1e-4, nesterov=True)
        o_dev = FusedSGD(dev_p, lr=1e-2, momentum=0.9, weight_decay=1e-4, nesterov=True)
    for it in range(4):
        for a, b in zip(ref_p, dev_p):
            gr = torch.randn(a.shape, generator=g)
            a.grad, b.grad = gr.clone(), gr.clone().cuda()
        o_ref.step(); o_dev.step()
    for a, b in zip(ref_p, dev_p):
        np.testing.assert_allclose(b.detach().cpu().numpy(), a.detach().numpy(), rtol=2e-6, atol=2e-7)


def test_confidence_mask_matches_oracle():
    from oracle.mean_teacher_ref import conf_mask_ref
    from uda_poseestimation_amd import utils as U
    hm = torch.rand(8, 16, 64, 64, generator=torch.Generator().manual_seed(2))
    hm[0, 0] = hm[0, 1]                                                   # an exact tie among the activations
    for ratio in (0.5, 0.25, 0.9):
        m_ref, act_ref, thr_ref = conf_mask_ref(hm, ratio)
        m, act, thr = U.confidence_mask(hm.cuda(), ratio)
        np.testing.assert_array_equal(act.cpu().numpy(), act_ref.numpy())
        assert float(thr) == thr_ref
        np.testing.assert_array_equal(m.cpu().numpy(), m_ref.numpy())


def test_warp_chain_matches_torchvision_restatement():
    from oracle.affine_ref import affine_nearest_ref, warp3_ref
    from uda_poseestimation_amd import synthetic, warp
    g = torch.Generator().manual_seed(3)
    x = torch.randn(6, 5, 64, 64, generator=g)
    ap = synthetic.aug_params(6, np.random.RandomState(5))
    ap[1][0][0], ap[1][1][0] = 8, -4                      # integer heat-map shifts after /ratio
    ap[1][0][1], ap[1][1][1] = 2, 6                       # half-pixel shifts: exact ties -> round-half-even
    y = warp.recon_heatmaps(x.cuda(), ap, 4.0).cpu()
    angle, (tx, ty), (sx, sy), sc = ap
    ref = torch.stack([warp3_ref(x[i], float(angle[i]), float(tx[i]), float(ty[i]), float(sx[i]), float(sy[i]), float(sc[i]), 4.0)
                       for i in range(6)])
    mism = (y != ref).float().mean().item()
    assert mism < 2e-3, mism                              # identical up to isolated nearest-neighbour ties ...
    # ... PROVABLY ties: every differing pixel's fp64 source coordinate lies within 1e-4 of a half-integer at some stage of the
    # chain, where two correct fp32 evaluations may round to different neighbours; no pixel differs anywhere else
    mism_px, exposed = (y != ref).any(dim=1), _tie_exposed(ap, 6, 64, 64, 4.0)
    print(f"re-warp vs the torchvision restatement: {int(mism_px.sum())} of {mism_px.numel()} pixels differ, all among the {int(exposed.sum())} tie-exposed ones")
    assert not bool((mism_px & ~exposed).any()), f"{int((mism_px & ~exposed).sum())} pixels differ from the oracle away from any rounding tie"
    # pure integer translation and identity are exact
    ident = warp.warp_chain(x.cuda(), warp.single_thetas(0.0, (0.0, 0.0), 1.0, (0.0, 0.0), 6, "cuda")).cpu()
    assert torch.equal(ident, x)
    sh = warp.warp_chain(x.cuda(), warp.single_thetas(0.0, (3.0, -2.0), 1.0, (0.0, 0.0), 6, "cuda")).cpu()
    assert torch.equal(sh[0], affine_nearest_ref(x[0], 0.0, [3.0, -2.0], 1.0, [0.0, 0.0]))
    # backward = transpose of the gather
    xd = x.cuda().requires_grad_(True)
    th = warp.recon_thetas(ap, 6, 4.0, "cuda")
    r = torch.randn(x.shape, generator=g)
    (warp.warp_chain(xd, th) * r.cuda()).sum().backward()
    xr = x.clone().requires_grad_(True)
    out = torch.stack([warp3_ref(xr[i], float(angle[i]), float(tx[i]), float(ty[i]), float(sx[i]), float(sy[i]), float(sc[i]), 4.0)
                       for i in range(6)])
    (out * r).sum().backward()
    # the device's backward is the exact transpose of the device's gather, the oracle's of the oracle's: the two gradients differ only where the
    # forwards differ - each differing output element moves one contribution between two input elements (and input elements that collect
    # several contributions may add them in another order: 1e-5 of the scale)
    gd, gr = xd.grad.cpu(), xr.grad
    n_diff = int(((gd - gr).abs() > 1e-5 * float(gr.abs().max())).sum())
    assert n_diff <= 2 * int((y != ref).sum()), (n_diff, int((y != ref).sum()))
    lhs = (warp.warp_chain(x.cuda(), th).double().cpu() * r.double()).sum().item()
    rhs = (x.double() * gd.double()).sum().item()
    assert abs(lhs - rhs) <= 1e-6 * (x.abs().double().sum().item()) + 1e-9, (lhs, rhs)      # <warp x, r> == <x, warp^T r>


@pytest.mark.parametrize("cs,ss", [((150, 106), (90, 122)), ((64, 200), (64, 200)), ((33, 47), (256, 256))], ids=["150x106_90x122", "64x200", "33x47_256"])
def test_style_net_on_odd_sizes_matches_the_oracle(cs, ss):
    """The style pass on sizes off the loop's 256x256: odd maps through the encoder's ceil-mode pools (150 -> 75 -> 38 -> 19), content and style of
    different sizes, a non-square strip, a map smaller than the patch kernels' tiles - the default fp32-grade mode against the fp32 CPU oracle
    (same seeded weights) within north_star's 1e-3 of max, the output size following the reference's (x8 of the relu4_1 map)."""
    from seeded import fill_style_weights
    from oracle import style_ref
    from uda_poseestimation_amd.lib.models import Style_net
    vgg_r, dec_r = style_ref.make_vgg_ref(), style_ref.make_decoder_ref()
    for m_, seed in ((vgg_r, 11), (dec_r, 12), (Style_net.vgg, 11), (Style_net.decoder, 12)):
        fill_style_weights(m_, seed)
    Style_net.vgg.cuda(); Style_net.decoder.cuda()
    vgg31 = torch.nn.Sequential(*list(Style_net.vgg.children())[:31])
    vgg31_r = torch.nn.Sequential(*list(vgg_r.children())[:31])
    net = Style_net.Net(vgg31, Style_net.decoder).cuda().eval()
    g = torch.Generator().manual_seed(cs[0] + ss[1])
    content, style = torch.rand(2, 3, *cs, generator=g), torch.rand(2, 3, *ss, generator=g)
    with torch.no_grad():
        want = style_ref.style_forward_ref(vgg31_r, dec_r, content, style, 0.7)
        got = net(content.cuda(), style.cuda(), 0.7)[2]
    assert got.shape == want.shape
    err = (got.cpu() - want).abs().max().item() / want.abs().max().item()
    assert err <= 1e-3, err


def test_captured_loss_section_backward_is_right_on_every_replay():
    """The consistency branch of the loss section - re-warp (autograd), ConsLoss, backward to dL/dy_t - captured alone into a small
    hipGraph and replayed with eager device work between the replays: every replay equals the eager result.  (With the re-warp's
    backward clearing its output by hipMemsetAsync the third and later replays ran the clear AFTER the scatter on ROCm 7.2:
    tools/probe/graph_memset_order.py; the clears are kernels now, csrc/common.h pw_zero.)"""
    from uda_poseestimation_amd import synthetic, warp
    from uda_poseestimation_amd.lib.models.loss import ConsLoss
    torch.manual_seed(0)
    N, K, H = 4, 16, 32
    out_t = (torch.rand(N, K, H, H) * 0.3).cuda()
    tea = torch.rand(N, K, H, H).cuda()
    mask = (torch.rand(N, K) > 0.5).cuda()
    theta = warp.recon_thetas(synthetic.aug_params(N, np.random.RandomState(2)), N, 4.0, "cuda")
    con = ConsLoss()

    def section():
        y_t = out_t.detach().requires_grad_(True)
        (con(warp.warp_chain(y_t, theta), tea, tea_mask=mask) * 1.0).backward()
        return y_t.grad

    want = section().cpu()
    assert want.abs().max() > 0
    torch.cuda.synchronize()
    stream = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream, capture_error_mode="global"):
        d_t = section()
    for rep in range(6):
        with torch.cuda.stream(stream):
            g.replay()
        torch.cuda.synchronize()
        busy = (d_t * 2).abs().max().item()         # (eager allocations and launches between the replays)
        err = (d_t.cpu() - want).abs().max().item()
        assert err <= 1e-9 + 1e-4 * want.abs().max().item(), (rep, err, busy)


def test_style_net_matches_reference_golden(golden_dir):
    """A10-A13 against the reference's own outputs (tests/golden/style.npz): bf16 fast mode with its stated error, fp32 mode
    (the reference's precision) within 1e-3 * max, content / Gram style losses of the full forward, module-level helpers."""
    from seeded import fill_style_weights
    from uda_poseestimation_amd.lib.models import Style_net
    z = _g(golden_dir, "style.npz")
    fill_style_weights(Style_net.vgg, 11)
    fill_style_weights(Style_net.decoder, 12)
    Style_net.vgg.cuda(); Style_net.decoder.cuda()
    vgg31 = torch.nn.Sequential(*list(Style_net.vgg.children())[:31])
    net = Style_net.Net(vgg31, Style_net.decoder).cuda().eval()
    content, style = torch.from_numpy(z["content"]).cuda(), torch.from_numpy(z["style"]).cuda()

    def err(a, b):
        return (a.cpu() - torch.from_numpy(b)).abs().max().item() / np.abs(b).max()
    # ---- bf16 (the fast mode; the default is the fp32-grade 'f16x2', tests/test_gpu_f16x2.py): the loop's [2]; losses are not
    # computed (NaN, never a silent zero)
    assert net.precision == "f16x2" and net.compute_losses is False
    net.precision = "bf16"
    with torch.no_grad():
        lc, ls, g_t = net(content, style, float(z["alpha"]))
        feat = net.encode(content)
        feats = net.encode_with_intermediate(style)
    assert [f.shape[1] for f in feats] == [64, 128, 256, 512]
    assert torch.isnan(lc) and torch.isnan(ls)
    e_feat, e_g = err(feat, z["feat"]), err(g_t, z["g_t"])
    print(f"style bf16: relu4_1 err {e_feat:.2e} * max, g_t err {e_g:.2e} * max")
    assert e_feat <= 4e-2 and e_g <= 8e-2      # 10 / 19 un-normalised bf16 conv layers; the reference is fp32
    with pytest.raises(AssertionError):
        net(content, style, 1.5)
    lo, hi = torch.tensor([-0.5, -0.4, -0.3]).cuda(), torch.tensor([0.5, 0.6, 0.7]).cuda()
    g_c = net(content, style, 0.6, clamp=(lo, hi))[2]
    ref_c = torch.maximum(torch.minimum(g_t.permute(0, 2, 3, 1), hi), lo).permute(0, 3, 1, 2)
    assert torch.allclose(g_c, ref_c, atol=1e-6)
    # ---- fp32 (the reference's precision): exact fp32 MFMA convolutions, fp32 AdaIN; full forward with both losses
    net.precision, net.compute_losses = "fp32", True
    with torch.no_grad():
        lc, ls, g32 = net(content, style, float(z["alpha"]))
        feat32 = net.encode(content)
    e_feat, e_g = err(feat32, z["feat"]), err(g32, z["g_t"])
    print(f"style fp32: relu4_1 err {e_feat:.2e} * max, g_t err {e_g:.2e} * max; loss_c {float(lc):.6f} (ref {float(z['loss_c']):.6f}) "
          f"loss_s {float(ls):.6e} (ref {float(z['loss_s']):.6e})")
    assert e_feat <= 1e-3 and e_g <= 1e-3
    assert abs(float(lc) - float(z["loss_c"])) <= 1e-3 * float(z["loss_c"])
    assert abs(float(ls) - float(z["loss_s"])) <= 1e-3 * float(z["loss_s"])
    g_c32 = net(content, style, 0.6, clamp=(lo, hi))
    assert torch.allclose(g_c32[2], torch.maximum(torch.minimum(g32.permute(0, 2, 3, 1), hi), lo).permute(0, 3, 1, 2), atol=1e-6)
    assert abs(float(g_c32[0]) - float(lc)) <= 1e-6 * abs(float(lc))           # the losses see the UNclamped g_t (Style_net.py:170)
    # bf16 mode computes the losses too (on its own, coarser features)
    net.precision = "bf16"
    with torch.no_grad():
        lc16, ls16, _ = net(content, style, float(z["alpha"]))
    assert abs(float(lc16) - float(z["loss_c"])) <= 0.15 * float(z["loss_c"]) and abs(float(ls16) - float(z["loss_s"])) <= 0.3 * float(z["loss_s"])
    net.compute_losses = False
    # ---- module-level helpers keep the reference's NCHW fp32 API (fp32 inside)
    c, s = torch.from_numpy(z["c"]).cuda(), torch.from_numpy(z["s"]).cuda()
    m, sd = Style_net.calc_mean_std(c)
    np.testing.assert_allclose(m.cpu().numpy(), z["mean"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(sd.cpu().numpy(), z["std"], rtol=1e-4)
    assert err(Style_net.adain(c, s), z["adain"]) <= 1e-5
    assert err(Style_net.gram_matrix(torch.from_numpy(z["feat"]).cuda()), z["gram"]) <= 1e-5
    with pytest.raises(AssertionError):
        Style_net.adain(c, s[:, :100])


def test_cons_loss_valid_mask_matches_reference_golden(golden_dir):
    """ConsLoss(valid_mask=) (loss.py:129-130): forward against the reference's own outputs, backward against autograd of the
    CPU oracle."""
    from oracle.losses_ref import cons_loss_ref
    from uda_poseestimation_amd.lib.models.loss import ConsLoss
    z = _g(golden_dir, "losses.npz")
    pred, gt, mask, valid = (torch.from_numpy(z[k]) for k in ("pred", "gt", "mask", "valid"))
    for tm, key in ((mask, "cons_valid"), (None, "cons_valid_only")):
        p_d = pred.cuda().requires_grad_(True)
        out = ConsLoss()(p_d, gt.cuda(), valid_mask=valid.cuda(), tea_mask=None if tm is None else tm.cuda())
        np.testing.assert_allclose(out.item(), float(z[key]), rtol=1e-6)
        (out * 3.0).backward()
        p_r = pred.clone().requires_grad_(True)
        (cons_loss_ref(p_r, gt, valid_mask=valid, tea_mask=tm) * 3.0).backward()
        np.testing.assert_allclose(p_d.grad.cpu().numpy(), p_r.grad.numpy(), rtol=1e-5, atol=1e-10)
    with pytest.raises(IndexError):
        ConsLoss()(pred.cuda(), gt.cuda(), valid_mask=valid[:, :10].cuda())


def test_mean_teacher_step_matches_cpu_oracle():
    """One whole step (train_human.py:326-440 order) on a small PoseResNet: losses, mask, EMA and Adam vs oracle/step_ref."""
    from oracle.pose_resnet_ref import PoseResNetRef
    from oracle.step_ref import train_step_ref
    from uda_poseestimation_amd import synthetic
    from uda_poseestimation_amd.engine import GraphedTrainStep, MeanTeacherTrainer
    import uda_poseestimation_amd.lib.models.pose_resnet as pr
    layers, K, N, S = [1, 1, 1, 1], 16, 4, 128
    torch.manual_seed(0)
    ref_s, ref_t = PoseResNetRef(layers, K), PoseResNetRef(layers, K)
    stu = pr._pose_resnet("t", K, pr.Bottleneck_default, layers, False, False)
    tea = pr._pose_resnet("t", K, pr.Bottleneck_default, layers, False, False)
    stu.load_state_dict(ref_s.state_dict())
    stu, tea = stu.cuda(), tea.cuda()
    trainer = MeanTeacherTrainer(stu, tea, image_size=S, heatmap_size=S // 4)      # EMA ctor copies student -> teacher
    ref_t.load_state_dict(ref_s.state_dict())
    for a, b in zip(tea.parameters(), ref_t.parameters()):
        assert torch.equal(a.detach().cpu(), b.detach())
    b = synthetic.mean_teacher_batch(N, num_keypoints=K, image_size=S, heatmap_size=S // 4, seed=7)
    g = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}
    w0 = [p.detach().clone() for p in ref_s.parameters()]
    out = trainer.train_step(g["x_s"], g["label_s"], g["weight_s"], g["x_t_stu"], g["x_t_tea"], g["aug_param_stu"], g["aug_param_tea"],
                             with_accuracy=True)
    opt = torch.optim.Adam(ref_s.parameters(), lr=1e-4)
    ref = train_step_ref(ref_s, ref_t, opt, b["x_s"], b["label_s"], b["weight_s"], b["x_t_stu"], b["x_t_tea"], b["aug_param_stu"],
                         b["aug_param_tea"], ratio=4.0)
    assert abs(float(out["loss_s"]) - float(ref["loss_s"])) <= 2e-2 * float(ref["loss_s"])
    assert abs(float(out["loss_c"]) - float(ref["loss_c"])) <= 6e-2 * float(ref["loss_c"]) + 1e-6
    # Adam moved every weight by ~lr in the same direction as the oracle for the overwhelming majority of entries
    agree, total = 0, 0
    for p_dev, p_ref, p0 in zip(stu.parameters(), ref_s.parameters(), w0):
        d_dev, d_ref = p_dev.detach().cpu() - p0, p_ref.detach() - p0
        sel = d_ref.abs() > 5e-5
        agree += int((torch.sign(d_dev[sel]) == torch.sign(d_ref[sel])).sum())
        total += int(sel.sum())
    # (first Adam step = lr * sign(grad): the fraction below is the sign agreement of the bf16 gradients with the fp32
    # oracle's, consistent with their measured cosine similarity of ~0.95-0.98, see tests/test_gpu_net.py)
    assert agree / max(total, 1) > 0.8, agree / max(total, 1)
    # EMA: teacher = 0.999*teacher + 0.001*student, on the device student
    for p_t, p_s, p0 in zip(tea.parameters(), stu.parameters(), w0):
        exp = p0.cuda().mul(0.999).add(p_s.detach() * (1.0 - 0.999))
        assert torch.equal(p_t.detach(), exp)
    # the captured (hipGraph) step reproduces the eager step on identical state
    torch.manual_seed(0)
    s2 = pr._pose_resnet("t", K, pr.Bottleneck_default, layers, False, False)
    t2 = pr._pose_resnet("t", K, pr.Bottleneck_default, layers, False, False)
    s3 = pr._pose_resnet("t", K, pr.Bottleneck_default, layers, False, False)
    t3 = pr._pose_resnet("t", K, pr.Bottleneck_default, layers, False, False)
    s3.load_state_dict(s2.state_dict())
    tr2 = MeanTeacherTrainer(s2.cuda(), t2.cuda(), image_size=S, heatmap_size=S // 4)
    tr3 = MeanTeacherTrainer(s3.cuda(), t3.cuda(), image_size=S, heatmap_size=S // 4)
    args = (g["x_s"], g["label_s"], g["weight_s"], g["x_t_stu"], g["x_t_tea"], g["aug_param_stu"], g["aug_param_tea"])
    gs = GraphedTrainStep(tr2, *args, warmup=2)
    for _ in range(2):
        tr3.train_step(*args)
    o2 = gs.step(*args)
    o3 = tr3.train_step(*args)
    assert abs(float(o2["loss_all"]) - float(o3["loss_all"])) <= 2e-3 * abs(float(o3["loss_all"]))
    # (the data-parallel split-graph form is compared with its eager twin from identical state in
    # tests/test_gpu_steps.py::test_captured_steps_equal_eager_steps_from_identical_state_with_varying_batches: a looser check of it here could not fail)


@pytest.mark.parametrize("N,K,S", [(1, 21, 96), (3, 17, 160), (5, 18, 64)], ids=["n1_k21_96", "n3_k17_160", "n5_k18_64"])
def test_mean_teacher_step_on_ragged_shapes_matches_cpu_oracle(N, K, S):
    """The whole step off the benchmark's grid - one image (BatchNorm over 3x3 = 9 values at layer4), key-point counts 17 / 18 / 21, 5x5 and 2x2
    layer4 maps, N*K not a multiple of anything: losses and the confidence mask against oracle/step_ref, the EMA bit for bit, and the
    captured step against the eager one from identical state."""
    from oracle.pose_resnet_ref import PoseResNetRef
    from oracle.step_ref import train_step_ref
    from uda_poseestimation_amd import synthetic
    from uda_poseestimation_amd.engine import GraphedTrainStep, MeanTeacherTrainer
    import uda_poseestimation_amd.lib.models.pose_resnet as pr
    layers = [1, 1, 1, 1]
    torch.manual_seed(N + K)
    ref_s, ref_t = PoseResNetRef(layers, K), PoseResNetRef(layers, K)
    mk = lambda: pr._pose_resnet("t", K, pr.Bottleneck_default, layers, False, False)
    stu, tea = mk(), mk()
    stu.load_state_dict(ref_s.state_dict())
    stu, tea = stu.cuda(), tea.cuda()
    trainer = MeanTeacherTrainer(stu, tea, image_size=S, heatmap_size=S // 4)
    ref_t.load_state_dict(ref_s.state_dict())
    b = synthetic.mean_teacher_batch(N, num_keypoints=K, image_size=S, heatmap_size=S // 4, seed=N * 10 + K)
    g = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}
    args = (g["x_s"], g["label_s"], g["weight_s"], g["x_t_stu"], g["x_t_tea"], g["aug_param_stu"], g["aug_param_tea"])
    w0 = [p.detach().clone() for p in ref_s.parameters()]
    out = trainer.train_step(*args)
    opt = torch.optim.Adam(ref_s.parameters(), lr=1e-4)
    ref = train_step_ref(ref_s, ref_t, opt, b["x_s"], b["label_s"], b["weight_s"], b["x_t_stu"], b["x_t_tea"], b["aug_param_stu"],
                         b["aug_param_tea"], ratio=4.0)
    assert out["y_s"].shape == (N, K, S // 4, S // 4)
    assert abs(float(out["loss_s"]) - float(ref["loss_s"])) <= 3e-2 * float(ref["loss_s"])
    assert abs(float(out["loss_c"]) - float(ref["loss_c"])) <= 1e-1 * float(ref["loss_c"]) + 1e-6
    assert out["tea_mask"].shape == (N, K) and int(out["tea_mask"].sum()) == N * K - int(0.5 * N * K)
    for p_t, p_s, p0 in zip(tea.parameters(), stu.parameters(), w0):
        assert torch.equal(p_t.detach(), p0.cuda().mul(0.999).add(p_s.detach() * (1.0 - 0.999)))
    s2, t2, s3, t3 = mk(), mk(), mk(), mk()
    s3.load_state_dict(s2.state_dict())
    tr2 = MeanTeacherTrainer(s2.cuda(), t2.cuda(), image_size=S, heatmap_size=S // 4)
    tr3 = MeanTeacherTrainer(s3.cuda(), t3.cuda(), image_size=S, heatmap_size=S // 4)
    gs = GraphedTrainStep(tr2, *args, warmup=1)
    tr3.train_step(*args)
    for _ in range(3):
        o2, o3 = gs.step(*args), tr3.train_step(*args)
        assert torch.isfinite(o2["loss_all"]) and abs(float(o2["loss_all"]) - float(o3["loss_all"])) <= 5e-3 * abs(float(o3["loss_all"]))


def test_reference_precision_mix_step_on_a_ragged_shape():
    """The reference's precision mix (fp16 student under the device-side loss scaler, fp32-grade f16x2 teacher) at N = 3, K = 17, 160x160:
    source loss against the fp32 oracle, the teacher-side quantities (mask; consistency loss within the student's fp16 error), no skipped
    step, and the captured step against the eager one."""
    from oracle.pose_resnet_ref import PoseResNetRef
    from oracle.step_ref import train_step_ref
    from uda_poseestimation_amd import synthetic
    from uda_poseestimation_amd.engine import GraphedTrainStep, MeanTeacherTrainer
    import uda_poseestimation_amd.lib.models.pose_resnet as pr
    layers, N, K, S = [1, 1, 1, 1], 3, 17, 160
    torch.manual_seed(3)
    ref_s, ref_t = PoseResNetRef(layers, K), PoseResNetRef(layers, K)
    mk = lambda: pr._pose_resnet("t", K, pr.Bottleneck_default, layers, False, False)
    stu = mk()
    stu.load_state_dict(ref_s.state_dict())
    tea = mk()
    trainer = MeanTeacherTrainer(stu.cuda(), tea.cuda(), image_size=S, heatmap_size=S // 4, precision="reference")
    ref_t.load_state_dict(ref_s.state_dict())
    b = synthetic.mean_teacher_batch(N, num_keypoints=K, image_size=S, heatmap_size=S // 4, seed=77)
    g = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}
    args = (g["x_s"], g["label_s"], g["weight_s"], g["x_t_stu"], g["x_t_tea"], g["aug_param_stu"], g["aug_param_tea"])
    w0 = [p.detach().clone() for p in stu.parameters()]
    out = trainer.train_step(*args)
    opt = torch.optim.Adam(ref_s.parameters(), lr=1e-4)
    ref = train_step_ref(ref_s, ref_t, opt, b["x_s"], b["label_s"], b["weight_s"], b["x_t_stu"], b["x_t_tea"], b["aug_param_stu"],
                         b["aug_param_tea"], ratio=4.0)
    assert abs(float(out["loss_s"]) - float(ref["loss_s"])) <= 1e-2 * float(ref["loss_s"])
    assert abs(float(out["loss_c"]) - float(ref["loss_c"])) <= 5e-2 * float(ref["loss_c"]) + 1e-6
    assert torch.equal(out["tea_mask"].cpu(), ref["tea_mask"].bool()) if "tea_mask" in ref else True
    assert any(not torch.equal(p.detach(), q) for p, q in zip(stu.parameters(), w0))          # (the scaler did not skip the step)
    s2, t2, s3, t3 = mk(), mk(), mk(), mk()
    s3.load_state_dict(s2.state_dict())
    tr2 = MeanTeacherTrainer(s2.cuda(), t2.cuda(), image_size=S, heatmap_size=S // 4, precision="reference")
    tr3 = MeanTeacherTrainer(s3.cuda(), t3.cuda(), image_size=S, heatmap_size=S // 4, precision="reference")
    gs = GraphedTrainStep(tr2, *args, warmup=1)
    tr3.train_step(*args)
    for _ in range(3):
        o2, o3 = gs.step(*args), tr3.train_step(*args)
        assert torch.isfinite(o2["loss_all"]) and abs(float(o2["loss_all"]) - float(o3["loss_all"])) <= 5e-3 * abs(float(o3["loss_all"]))


def test_occlusion_matches_oracle():
    """A16 (train_human.py:374-412): same host draws, same boxes, images identical up to isolated nearest-neighbour ties."""
    from oracle.occlusion_ref import occlude_ref
    from uda_poseestimation_amd import synthetic, warp
    g = torch.Generator().manual_seed(8)
    B, K, S = 6, 16, 128
    x = synthetic.images(B, S, 5)
    recon = torch.rand(B, K, S // 4, S // 4, generator=g) * 0.8
    recon[0, 3, 10, 12] = 0.95; recon[0, 7, 2, 30] = 0.99      # confident key-points (>= 0.9) on some samples only
    recon[2, 1, 31, 0] = 0.97; recon[4, 15, 0, 0] = 1.5; recon[5, 9, 20, 20] = 0.91
    ap = synthetic.aug_params(B, np.random.RandomState(3))
    ref, chosen_ref = occlude_ref(x, recon, ap, 4.0, S, 0.7, 0.9, 10, np.random.RandomState(11))
    out, chosen = warp.occlude_keypoints(x.cuda(), recon.cuda(), ap, 4.0, S, 0.7, 0.9, 10, np.random.RandomState(11))
    assert chosen == chosen_ref and len(chosen) >= 2
    out = out.cpu()
    untouched = [b for b in range(B) if b not in chosen]
    assert torch.equal(out[untouched], x[untouched])
    mism = (out[chosen] != ref[chosen]).float().mean().item()
    assert mism < 5e-3, mism
    # disabled exactly like `--occlude-rate -1`... and rate 0 selects nothing
    o2, c2 = warp.occlude_keypoints(x.cuda(), recon.cuda(), ap, 4.0, S, 0.0, 0.9, 10, np.random.RandomState(1))
    assert c2 == [] or all(np.random.RandomState(1).rand() <= 0.0 for _ in c2)


def test_device_side_occlusion_decisions_match_oracle():
    """The same occlusion with the decisions taken on the device from four uniform draws per sample (no read-back): the selected
    samples, boxes and patch origins follow the oracle fed with the same draws; unselected samples come back bit-identical."""
    from oracle.occlusion_ref import occlude_from_uniforms_ref
    from uda_poseestimation_amd import synthetic, warp
    g = torch.Generator().manual_seed(8)
    B, K, S = 12, 16, 128
    x = synthetic.images(B, S, 5)
    recon = torch.rand(B, K, S // 4, S // 4, generator=g) * 0.8
    for b, k, r, c, v in ((0, 3, 10, 12, 0.95), (0, 7, 2, 30, 0.99), (2, 1, 31, 0, 0.97), (4, 15, 0, 0, 1.5), (5, 9, 20, 20, 0.91),
                          (7, 0, 16, 16, 0.93), (7, 5, 1, 1, 0.92), (7, 11, 30, 30, 0.96), (9, 2, 8, 25, 0.9), (11, 14, 31, 31, 2.0)):
        recon[b, k, r, c] = v
    ap = synthetic.aug_params(B, np.random.RandomState(3))
    u = torch.from_numpy(np.random.RandomState(5).rand(B, 4).astype(np.float32))
    u[7, 0] = 0.0; u[7, 1] = 0.999999; u[11, 0] = 0.1; u[11, 2] = 0.999999; u[11, 3] = 0.0     # edge draws: last candidate, last origin
    ref, chosen_ref = occlude_from_uniforms_ref(x, recon, ap, 4.0, S, 0.7, 0.9, 10, u)
    fwd = warp.recon_thetas(ap, B, 4.0, "cuda")
    back = warp.occlusion_back_thetas(ap, B, 4.0, "cuda")
    out, apply = warp.occlude_keypoints_device(x.cuda(), recon.cuda(), fwd, back, u.cuda(), 4.0, S, 0.7, 0.9, 10)
    chosen = [int(i) for i in torch.nonzero(apply.cpu()).flatten()]
    assert chosen == chosen_ref and len(chosen) >= 3 and len(chosen) < B
    out = out.cpu()
    untouched = [b for b in range(B) if b not in chosen]
    assert torch.equal(out[untouched], x[untouched])
    mism = (out[chosen] != ref[chosen]).float().mean().item()
    assert mism < 5e-3, mism
    # ... and identical to the host-decision path when that is fed the same picks (same warps, same paste)
    from oracle.occlusion_ref import _UniformDraws

    class _Seq:     # the host path draws sequentially over the qualifying samples: serve each sample's four numbers in turn
        def __init__(self, u, qualifying):
            self.it = iter(qualifying); self.u = u; self.cur = None
        def rand(self):
            self.cur = _UniformDraws(self.u[next(self.it)].tolist()); return self.cur.rand()
        def choice(self, c):
            return self.cur.choice(c)
        def randint(self, m):
            return self.cur.randint(m)
    qualifying = [b for b in range(B) if bool((recon[b].amax(dim=(1, 2)) >= 0.9).any())]
    host, chosen_h = warp.occlude_keypoints(x.cuda(), recon.cuda(), ap, 4.0, S, 0.7, 0.9, 10, _Seq(u, qualifying))
    assert chosen_h == chosen and torch.equal(host.cpu(), out)
    # rate <= -1 / no confident key point: nothing selected
    out0, apply0 = warp.occlude_keypoints_device(x.cuda(), (recon * 0.1).cuda(), fwd, back, u.cuda(), 4.0, S, 0.7, 0.9, 10)
    assert int(apply0.sum()) == 0 and torch.equal(out0.cpu(), x)


def test_style_and_occlusion_step_runs_config2():
    """BASELINE.json configs[2]: the step with AdaIN s2t/t2s style passes (seeded random VGG/decoder) and occlusion."""
    from seeded import fill_style_weights
    from uda_poseestimation_amd import synthetic
    from uda_poseestimation_amd.engine import MeanTeacherTrainer
    from uda_poseestimation_amd.lib.models import Style_net
    import uda_poseestimation_amd.lib.models.pose_resnet as pr
    fill_style_weights(Style_net.vgg, 11)
    fill_style_weights(Style_net.decoder, 12)
    Style_net.vgg.cuda(); Style_net.decoder.cuda()
    net = Style_net.Net(torch.nn.Sequential(*list(Style_net.vgg.children())[:31]), Style_net.decoder).cuda()
    layers, K, N, S = [1, 1, 1, 1], 16, 4, 128
    torch.manual_seed(0)
    stu = pr._pose_resnet("t", K, pr.Bottleneck_default, layers, False, False).cuda()
    tea = pr._pose_resnet("t", K, pr.Bottleneck_default, layers, False, False).cuda()
    lo = torch.tensor([-2.1179, -2.0357, -1.8044]).cuda()
    hi = torch.tensor([2.2489, 2.4285, 2.64]).cuda()
    tr = MeanTeacherTrainer(stu, tea, image_size=S, heatmap_size=S // 4, style_net=net, recover=(lo, hi), s2t_freq=1.0, t2s_freq=1.0,
                            s2t_alpha=(0.5, 0.5), t2s_alpha=(0.5, 0.5), rng=np.random.RandomState(0), occlude_rate=0.5, occlude_thresh=-1e9,
                            occlude_size=10)
    b = synthetic.mean_teacher_batch(N, num_keypoints=K, image_size=S, heatmap_size=S // 4, seed=2)
    g = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}
    w0 = [p.detach().clone() for p in stu.parameters()]
    for _ in range(2):
        out = tr.train_step(g["x_s"], g["label_s"], g["weight_s"], g["x_t_stu"], g["x_t_tea"], g["aug_param_stu"], g["aug_param_tea"])
    assert all(torch.isfinite(out[k]).all() for k in ("loss_all", "loss_s", "loss_c"))
    assert any(not torch.equal(a, p.detach()) for a, p in zip(w0, stu.parameters()))
    # the styled source image stays inside the recover clamp
    xs = net(g["x_s"], g["x_t_tea"], 0.5, clamp=(lo, hi))[2]
    assert float((xs - hi.view(1, 3, 1, 1)).max()) <= 1e-6 and float((lo.view(1, 3, 1, 1) - xs).max()) <= 1e-6


def test_animal_config_shapes_384_k18():
    """BASELINE.json configs[4] shapes: K=18, 384x384 -> 96x96 heat-maps, float sigma 1.0, animal clamp constants."""
    from oracle.pose_resnet_ref import PoseResNetRef
    from oracle.bf16_emulation import forward_bf16_emulated
    import uda_poseestimation_amd.lib.models.pose_resnet as pr
    from uda_poseestimation_amd import utils as U
    torch.manual_seed(0)
    ref = PoseResNetRef([1, 1, 1, 1], 18).train()
    net = pr._pose_resnet("t", 18, pr.Bottleneck_default, [1, 1, 1, 1], False, False)
    net.load_state_dict(ref.state_dict())
    net = net.cuda().train()
    x = torch.randn(2, 3, 384, 384, generator=torch.Generator().manual_seed(4))
    with torch.no_grad():
        y = net(x.cuda())
        y_emu = forward_bf16_emulated(ref, x)
        y_ref = ref(x)
    assert tuple(y.shape) == (2, 18, 96, 96)
    noise = (y_emu - y_ref).abs().max().item()
    assert (y.cpu() - y_ref).abs().max().item() <= 2.0 * noise + 2e-3 * y_ref.abs().max().item()
    r = U.rectify(y, 1.0)
    assert tuple(r.shape) == (2, 18, 96, 96) and float(r.max()) == 1.0
    net.precision = "fp32"
    with torch.no_grad():
        y32 = net(x.cuda())
    assert (y32.cpu() - y_ref).abs().max().item() < 1e-4


def test_mean_teacher_step_with_two_teacher_views_matches_oracle():
    """`--k 2` (train_human.py:358-372): two teacher views per step, each forwarded and re-warped with its own aug_param, the re-warped
    maps averaged per sample (`torch.mean(recons, dim=0)`) - udapose_mean_views against torch.mean bit for bit, and the whole step in
    the reference's precision mix (fp32-grade teacher) against oracle.step_ref.train_step_ref with view LISTS: mask element for
    element, both losses."""
    from oracle.pose_resnet_ref import PoseResNetRef
    from oracle.step_ref import train_step_ref
    from uda_poseestimation_amd import synthetic, warp
    from uda_poseestimation_amd.engine import MeanTeacherTrainer
    import uda_poseestimation_amd.lib.models.pose_resnet as pr
    vs = [torch.randn(3, 16, 32, 32, device="cuda") * (10.0 ** (i - 1)) for i in range(3)]
    for k in (2, 3):
        # (the reference builds `recons` with torch.zeros(k, ...) - a CPU tensor - so its mean is ATen's CPU mean: sum in view order, divided by k)
        assert torch.equal(warp.mean_views(vs[:k]).cpu(), torch.mean(torch.stack([v.cpu() for v in vs[:k]]), dim=0))
    assert warp.mean_views(vs[:1]) is vs[0]
    layers, K, N, S = [1, 1, 1, 1], 16, 4, 128
    torch.manual_seed(3)
    ref_s, ref_t = PoseResNetRef(layers, K), PoseResNetRef(layers, K)
    stu = pr._pose_resnet("t", K, pr.Bottleneck_default, layers, False, False)
    tea = pr._pose_resnet("t", K, pr.Bottleneck_default, layers, False, False)
    stu.load_state_dict(ref_s.state_dict())
    trainer = MeanTeacherTrainer(stu.cuda(), tea.cuda(), image_size=S, heatmap_size=S // 4, precision="reference")
    ref_t.load_state_dict(ref_s.state_dict())
    b = synthetic.mean_teacher_batch(N, num_keypoints=K, image_size=S, heatmap_size=S // 4, seed=7)
    b2 = synthetic.mean_teacher_batch(N, num_keypoints=K, image_size=S, heatmap_size=S // 4, seed=8)
    g = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}
    views, aps = [b["x_t_tea"], b2["x_t_tea"]], [b["aug_param_tea"], b2["aug_param_tea"]]
    out = trainer.train_step(g["x_s"], g["label_s"], g["weight_s"], g["x_t_stu"], [v.cuda() for v in views], g["aug_param_stu"], aps)
    opt = torch.optim.Adam(ref_s.parameters(), lr=1e-4)
    ref = train_step_ref(ref_s, ref_t, opt, b["x_s"], b["label_s"], b["weight_s"], b["x_t_stu"], views, b["aug_param_stu"], aps, ratio=4.0)
    assert tea._last_hd.precision == "f16x2" and stu._last_hd.precision == "fp16"
    mask_dev, mask_ref = out["tea_mask"].cpu().bool(), ref["tea_mask"].bool()
    assert (mask_dev != mask_ref).sum().item() <= 1, (mask_dev != mask_ref).sum().item()       # (a confidence within rounding of the k-th value may flip)
    assert abs(float(out["loss_s"]) - float(ref["loss_s"])) <= 5e-3 * float(ref["loss_s"])
    assert abs(float(out["loss_c"]) - float(ref["loss_c"])) <= 2e-2 * float(ref["loss_c"]) + 1e-7
    # ... and it is not the k = 1 step: the one-view oracle selects other key points (the mask the device reproduced above is the two-view one)
    torch.manual_seed(3)
    r_s, r_t = PoseResNetRef(layers, K), PoseResNetRef(layers, K)
    r_t.load_state_dict(r_s.state_dict())
    one = train_step_ref(r_s, r_t, torch.optim.Adam(r_s.parameters(), lr=1e-4), b["x_s"], b["label_s"], b["weight_s"], b["x_t_stu"], b["x_t_tea"],
                         b["aug_param_stu"], b["aug_param_tea"], ratio=4.0)
    assert (one["tea_mask"].bool() != mask_ref).sum().item() >= 2 and not torch.equal(one["y_t_tea_recon"], ref["y_t_tea_recon"])


def test_mean_teacher_step_animal_config_k18_float_sigma():
    """A whole three-stream step at the configs[4] shape family (K = 18 -> a parameter count that is not a multiple of 4, float
    sigma 1.0, 192x192 -> 48x48 maps): losses against oracle/step_ref, the two per-pass gradient buffers summed, replay of the
    captured step equal to the eager one."""
    from oracle.pose_resnet_ref import PoseResNetRef
    from oracle.step_ref import train_step_ref
    from uda_poseestimation_amd import synthetic
    from uda_poseestimation_amd.engine import GraphedTrainStep, MeanTeacherTrainer
    import uda_poseestimation_amd.lib.models.pose_resnet as pr
    layers, K, N, S = [1, 1, 1, 1], 18, 3, 192
    torch.manual_seed(1)
    ref_s, ref_t = PoseResNetRef(layers, K), PoseResNetRef(layers, K)
    nets = []
    for _ in range(2):
        stu = pr._pose_resnet("t", K, pr.Bottleneck_default, layers, False, False)
        tea = pr._pose_resnet("t", K, pr.Bottleneck_default, layers, False, False)
        stu.load_state_dict(ref_s.state_dict())
        nets.append(MeanTeacherTrainer(stu.cuda(), tea.cuda(), sigma=1.0, image_size=S, heatmap_size=S // 4))
    assert sum(p.numel() for p in nets[0].student.parameters()) % 4 != 0
    ref_t.load_state_dict(ref_s.state_dict())
    b = synthetic.mean_teacher_batch(N, num_keypoints=K, image_size=S, heatmap_size=S // 4, sigma=1.0, seed=11)
    g = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}
    args = (g["x_s"], g["label_s"], g["weight_s"], g["x_t_stu"], g["x_t_tea"], g["aug_param_stu"], g["aug_param_tea"])
    out = nets[0].train_step(*args)
    opt = torch.optim.Adam(ref_s.parameters(), lr=1e-4)
    ref = train_step_ref(ref_s, ref_t, opt, b["x_s"], b["label_s"], b["weight_s"], b["x_t_stu"], b["x_t_tea"], b["aug_param_stu"],
                         b["aug_param_tea"], ratio=4.0, sigma=1.0)
    assert abs(float(out["loss_s"]) - float(ref["loss_s"])) <= 2e-2 * float(ref["loss_s"])
    assert abs(float(out["loss_c"]) - float(ref["loss_c"])) <= 8e-2 * float(ref["loss_c"]) + 1e-6
    gs = GraphedTrainStep(nets[1], *args, warmup=1)          # (one eager step inside: both trainers have now done one step)
    o_g = gs.step(*args)
    o_e = nets[0].train_step(*args)
    loss_g1 = float(o_g["loss_all"])                 # (step() returns the graph's static output tensors: read the value now)
    assert abs(loss_g1 - float(o_e["loss_all"])) <= 3e-3 * abs(float(o_e["loss_all"])) + 1e-7
    # host batches staged by prefetch() (pinned memory -> copy stream -> staging buffers -> static inputs): same step
    b2 = synthetic.mean_teacher_batch(N, num_keypoints=K, image_size=S, heatmap_size=S // 4, sigma=1.0, seed=12)
    host = {k: v.pin_memory() for k, v in b2.items() if torch.is_tensor(v)}
    gs.prefetch(host["x_s"], host["label_s"], host["weight_s"], host["x_t_stu"], host["x_t_tea"])
    o_g2 = gs.step(None, None, None, None, None, b2["aug_param_stu"], b2["aug_param_tea"])
    g2 = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b2.items()}
    o_e2 = nets[0].train_step(g2["x_s"], g2["label_s"], g2["weight_s"], g2["x_t_stu"], g2["x_t_tea"], g2["aug_param_stu"], g2["aug_param_tea"])
    assert abs(float(o_g2["loss_all"]) - float(o_e2["loss_all"])) <= 5e-3 * abs(float(o_e2["loss_all"])) + 1e-7
    assert abs(float(o_g2["loss_all"]) - loss_g1) > 1e-6                          # (it really was a different batch)


@pytest.mark.parametrize("arch,batch", [("pose_resnet50", 4), ("pose_resnet101", 8)])
def test_two_rank_step_on_one_gpu_gloo(arch, batch):
    """The data-parallel launch (one process per rank, three hipGraphs cut around the confidence all-gather and the gradient
    all-reduce, max-over-ranks timing) run for real with two ranks; both share cuda:0 and talk over gloo, because the
    test box has one GPU (bench.py's UDAPOSE_BENCH_SHARE_GPU hook).  Checks the launch contract and that the two ranks
    end with the same averaged gradient step (identical losses are not expected: every rank has its own shard).  With a spin-up
    phase: its length is rank 0's decision, shared after every step (rank-local clocks would let the ranks disagree by a step)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, UDAPOSE_BENCH_SHARE_GPU="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--spinup", "0.7", "--arch", arch, "--batch", str(batch), "--no-cpu-baseline", "--dp-form", "fixed"]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]          # rank 0 prints ONE JSON line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak" and d["config"]["global_batch"] == 2 * batch
    assert d["value"] > 0 and d["loss"] == d["loss"]
    assert "4 hipGraphs" in d["launch"]               # forwards | losses + backward part 1 | backward part 2 | Adam + EMA
    assert d["replicas_in_sync"] is True               # same averaged gradients -> bit-identical replicas
    assert d["rccl_ranks"] == 0                        # (gloo here: the one-GPU box cannot host two RCCL ranks)
    # per-rank diagnostics of a multi-rank line: both ranks' own ms per step and their exposed communication time
    assert len(d["rank_comm_exposed_ms"]) == 2 and d["rank_ms_per_step_min_max"][0] <= d["rank_ms_per_step_min_max"][1]


def test_two_rank_synced_gradient_is_the_mean_of_the_rank_gradients(tmp_path):
    """VERDICT r3 weak #4 / next #5b-c: `replicas_in_sync` only says both ranks applied the SAME update.  Here two ranks (sharing cuda:0,
    gloo) run one data-parallel forward / backward / gradient exchange on their own shards (backward cut after layer3, suffix and
    prefix buckets) and write the synchronised flat gradient; this process then runs the two shards' passes itself on identical
    weights - the mask threshold from the CONCATENATED [2N, K] confidences - and checks (b) the exchanged gradient = 1/2 (g0 + g1) to
    fp32 rounding, identical on both ranks, and (c) each rank's consistency mask = the global-batch k-th-value mask
    (train_human.py:429, 145-148: one process sees the whole batch there)."""
    import os, subprocess, sys
    import numpy as np
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tests", "helpers"))
    import dp_rank_worker as W
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(root, "tests", "helpers", "dp_rank_worker.py"), str(tmp_path)]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    dumps = [np.load(os.path.join(str(tmp_path), f"rank{r}.npz")) for r in range(2)]
    assert np.array_equal(dumps[0]["flat"], dumps[1]["flat"])                    # both ranks hold the same exchanged gradient
    # the same two passes in ONE process (no process group): forwards first, then the global threshold, then each shard's backward
    from uda_poseestimation_amd import warp
    from uda_poseestimation_amd import utils as mt
    from uda_poseestimation_amd.engine import MeanTeacherTrainer
    th = lambda ap: warp.recon_thetas(ap, 4, 4.0, "cuda")
    trs, sts = [], []
    for r in range(2):
        stu, tea = W.build()
        tr = MeanTeacherTrainer(stu, tea, image_size=128, heatmap_size=32)
        g = W.shard(r)
        st = tr._forward_part(g["x_s"], g["label_s"], g["weight_s"], g["x_t_stu"], [g["x_t_tea"]], th(g["aug_param_stu"]), [th(g["aug_param_tea"])])
        trs.append(tr); sts.append(st)
    gathered = torch.cat([st["activates"].reshape(-1) for st in sts])
    grads = []
    for r in range(2):
        res = trs[r]._loss_backward_part(sts[r], gathered)
        trs[r]._sync_grads()
        trs[r].student.finish_grads()
        torch.cuda.synchronize()
        grads.append(trs[r].student._flat_grad.detach().cpu().numpy().copy())
        # (c) the rank's mask is the mask of the concatenated confidences
        act = sts[r]["activates"]
        k = int(0.5 * gathered.numel())
        thr = torch.kthvalue(gathered, k)[0]
        assert np.array_equal(dumps[r]["mask"].astype(bool), (act > thr).cpu().numpy()), r
        assert np.array_equal(res["tea_mask"].cpu().numpy().astype(bool), dumps[r]["mask"].astype(bool))
    mean = (grads[0] + grads[1]) * np.float32(0.5)
    d = np.abs(dumps[0]["flat"] - mean)
    scale = np.abs(mean).max()
    assert scale > 0 and d.max() <= 2e-6 * scale, (d.max(), scale)               # fp32 SUM x 1/2 of the two ranks' buffers
    assert np.abs(grads[0] - grads[1]).max() > 1e-3 * scale                      # (the shards' gradients really differ)


def test_comm_bf16_pack_shard_mean_unpack_emulating_w_ranks_on_one_gpu():
    """udapose_comm_pack_bf16 -> (all-to-all by hand) -> udapose_comm_shard_mean -> (all-gather by hand) -> udapose_comm_unpack_bf16 for
    W = 2 and 8 emulated ranks on one GPU (the RCCL path refuses gloo, so W > 1 never ran: VERDICT r3 weak #3): odd n, n
    not a multiple of 8 W.  Expected = what the wire format defines: every contribution rounded to bf16 ONCE, added in fp32 in rank order,
    times 1/W, the mean rounded to bf16 once more for the all-gather (two roundings per averaged element in all), widened to fp32."""
    from uda_poseestimation_amd._hip import check, lib, ptr, stream
    L = lib()
    for W_, n in ((2, 1001), (8, 4099), (8, 64), (3, 7)):
        torch.manual_seed(n)
        gs = [(torch.randn(n, device="cuda") * (10.0 ** torch.randint(-3, 3, (n,), device="cuda").float())) for _ in range(W_)]
        m = (n + W_ - 1) // W_
        m = (m + 7) // 8 * 8                                   # engine.GradSync._reduce_bf16's padding
        send = [torch.full((W_ * m,), float("nan"), dtype=torch.bfloat16, device="cuda") for _ in range(W_)]
        for r in range(W_):
            check(L.udapose_comm_pack_bf16(stream(), ptr(gs[r]), n, ptr(send[r]), W_ * m), "pack")
            assert torch.equal(send[r][:n], gs[r].to(torch.bfloat16)) and not send[r][n:].float().abs().sum().item()      # one rounding, zero padding
        mine = []
        for r in range(W_):                                    # all_to_all_single: rank r receives shard r of every rank, in rank order
            recv = torch.cat([send[j][r * m:(r + 1) * m] for j in range(W_)])
            o = torch.empty(m, dtype=torch.bfloat16, device="cuda")
            check(L.udapose_comm_shard_mean(stream(), ptr(recv), W_, m, ptr(o)), "shard_mean")
            mine.append(o)
        gathered = torch.cat(mine)                             # all_gather_into_tensor
        outp = torch.empty(n, device="cuda")
        check(L.udapose_comm_unpack_bf16(stream(), ptr(gathered), ptr(outp), n), "unpack")
        acc = torch.zeros(n, device="cuda")
        for r in range(W_):
            acc = acc + gs[r].to(torch.bfloat16).float()
        want = (acc * (1.0 / W_)).to(torch.bfloat16).float()
        assert torch.equal(outp, want), (W_, n, (outp - want).abs().max().item())
        exact = torch.stack(gs).double().mean(0)
        # distance from the exact fp64 mean: every contribution rounded once (2^-9 of its size each) and the mean once more
        bound = (torch.stack(gs).double().abs().mean(0) + exact.abs()) * 2.0 ** -8
        assert bool(((outp.double() - exact).abs() <= bound + 1e-30).all())


def test_two_rank_config3_step_style_and_occlusion_captured():
    """BASELINE.json configs[3] in small (data parallel + style / occlusion was never run, even at two
    ranks): two ranks sharing cuda:0 over gloo, each with its shard, the style directions and the occlusion decisions inside
    the captured step, the four data-parallel graphs around the collectives; replicas stay bit-identical."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(UDAPOSE_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--config2", "--steps", "3", "--warmup", "1", "--spinup", "0",
           "--arch", "pose_resnet50", "--batch", "4", "--no-cpu-baseline", "--dp-form", "fixed"]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 8 and d["replicas_in_sync"] is True
    assert "style-transfer hipGraphs" in d["launch"] and "4 hipGraphs" in d["launch"]
    assert d["loss"] == d["loss"] and d["value"] > 0


def test_bench_gpus_flag_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher and WORLD_SIZE unset: the parent spawns two fresh rank processes before
    touching the GPU, relays rank 0's ONE JSON line and exits with the children's status (`--gpus` used to be
    ignored).  A --gpus / WORLD_SIZE mismatch fails fast."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(UDAPOSE_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--spinup", "0",
           "--arch", "pose_resnet50", "--batch", "4", "--no-cpu-baseline"]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 8 and d["replicas_in_sync"] is True
    # (no form named: both gloo forms - two buckets | one bucket - were timed during spin-up and the ranks agreed on one)
    assert d["dp_form"]["chosen"] in ("two_buckets_fp32", "one_bucket_fp32") and len(d["dp_form"]["ms_per_step_5_steps"]) == 2
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "1"], cwd=root,
                         env=dict(env, WORLD_SIZE="2", RANK="0"), capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and "WORLD_SIZE=2" in (bad.stdout + bad.stderr)


def test_one_rank_rccl_step():
    """The data-parallel code path on the REAL RCCL backend with a one-rank process group (bench.py's UDAPOSE_FORCE_DIST hook:
    the test box has one GPU, and RCCL refuses two ranks on one device): backend "nccl" initialised with a device id, the
    gradient all-reduce and the confidence all-gather issued eagerly between the three hipGraph replays while the RCCL watchdog
    thread is alive, process-group teardown.  The same step without a process group gives the same loss (world size 1: the
    collectives are identities)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = [os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1", "--spinup", "0", "--arch", "pose_resnet50", "--batch", "4",
              "--no-cpu-baseline"]
    res = {}
    rccl_env = {"UDAPOSE_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port())}
    for tag, extra, flags in (("rccl", rccl_env, ["--dp-form", "fixed", "--capture-comm"]), ("plain", {}, []),
                              ("rccl_bf16", dict(rccl_env, MASTER_PORT=str(_free_port())), ["--grad-comm", "bf16", "--capture-comm"]),
                              ("rccl_auto", dict(rccl_env, MASTER_PORT=str(_free_port())), ["--capture-comm"]),
                              ("rccl_eager", dict(rccl_env, MASTER_PORT=str(_free_port())), []),
                              ("rccl_fail", dict(rccl_env, MASTER_PORT=str(_free_port()), UDAPOSE_TEST_FAIL_CAPTURE="1"), ["--capture-comm"])):
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **extra)
        out = subprocess.run([sys.executable] + common + flags, cwd=root, env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, tag + out.stdout[-2000:] + out.stderr[-4000:]
        lines = [l for l in out.stdout.splitlines() if l.strip()]
        assert len(lines) == 1 and lines[0].startswith("{"), out.stdout[-2000:]     # ONE JSON line, RCCL's banner goes to stderr
        res[tag] = json.loads(lines[0])
    # round 5: over RCCL the collectives are captured INTO the step's graph (one launch per step, as on one rank without a group), and with
    # no form named on the command line the bench times the four forms during spin-up and reports its choice
    assert "1 hipGraph with the RCCL collectives captured" in res["rccl"]["launch"] and "1 hipGraph (" in res["plain"]["launch"]
    dpf = res["rccl_auto"]["dp_form"]
    assert dpf["chosen"] in dpf["ms_per_step_5_steps"] and len(dpf["ms_per_step_5_steps"]) == 4 and res["plain"]["dp_form"] is None
    assert res["rccl"]["dp_form"] is None and res["rccl_auto"]["loss"] == res["rccl_auto"]["loss"]
    assert dpf["collectives_captured"] is (not dpf["chosen"] == "two_buckets_bf16") and dpf["capture_fallback"] is None
    # the DEFAULT data-parallel run (what the driver's `--gpus N` line is): the same four forms timed with EAGER collectives between four graphs
    dpe = res["rccl_eager"]["dp_form"]
    assert dpe["collectives_captured"] is False and dpe["capture_fallback"] is None and len(dpe["ms_per_step_5_steps"]) == 4
    assert "hipGraphs" in res["rccl_eager"]["launch"] and "captured" not in res["rccl_eager"]["launch"]
    e_ = res["rccl_eager"]["loss"]
    assert e_ == e_ and abs(e_ - res["rccl_auto"]["loss"]) <= 2e-2 * abs(e_) + 1e-9
    # SCALE-day checklist: the line carries the wall time of communicator creation, of every form's capture and of the selection; an
    # injected capture failure makes every form fall back - collectively, by an all-reduced flag - to eager collectives between four graphs, and the
    # line says so
    su = res["rccl_auto"]["dp_setup_s"]
    assert su["communicator_s"] >= 0 and set(su["capture_s"]) == set(dpf["ms_per_step_5_steps"]) and su["selection_s"] > 0
    print("dp setup wall times (one-rank RCCL, PoseResNet-50 b=4):", su)
    ff = res["rccl_fail"]["dp_form"]
    assert ff["collectives_captured"] is False and ff["capture_fallback"] and all("injected" in v for v in ff["capture_fallback"].values())
    assert "hipGraphs" in res["rccl_fail"]["launch"] and "captured" not in res["rccl_fail"]["launch"]
    # (same number of steps as the run whose forms were captured - the selection runs 4 x 7 steps before the timed region -, eager collectives instead)
    f_, a_ = res["rccl_fail"]["loss"], res["rccl_auto"]["loss"]
    assert f_ == f_ and abs(f_ - a_) <= 2e-2 * abs(a_) + 1e-9, (f_, a_)
    assert res["rccl"]["n_gpus"] == 1 and res["rccl"]["value"] > 0
    assert res["rccl"]["rccl_ranks"] == 1 and res["plain"]["rccl_ranks"] == 0
    a, b = res["rccl"]["loss"], res["plain"]["loss"]
    assert a == a and abs(a - b) <= 1e-3 * abs(b) + 1e-9, (a, b)
    # one SCALE line diagnoses itself: the time the compute stream waited in the communication calls is reported
    # (the eager four-graph form - rccl_bf16 below: the bf16 suffix exchange cannot be captured - still reports what the compute stream waited)
    assert res["rccl"]["grad_comm"] in ("fp32", "bf16") and res["plain"]["grad_comm"] is None
    assert res["rccl_bf16"]["comm_exposed_ms_per_step"] is not None
    # gradient buckets in bf16 on the wire (all-to-all of shards, fp32 accumulation on the owner, all-gather): one bf16 rounding of the
    # gradients at world size 1 - the same training step to the precision of that rounding
    c = res["rccl_bf16"]["loss"]
    assert res["rccl_bf16"]["grad_comm"] == "bf16" and "4 hipGraphs" in res["rccl_bf16"]["launch"]
    # (the exchange itself is exact at world size 1 - bf16(bf16(g)) = bf16(g); what differs from the fp32 run is Adam on gradients rounded
    # to 8 bits, through a randomly initialised train-mode-BN network: measured 0.8 % on the loss after four steps)
    assert c == c and abs(c - b) <= 2e-2 * abs(b) + 1e-9, (c, b)


def test_validate_matches_cpu_oracle():
    """validate() (train_human.py:461-500): eval-mode forward, device decode + PCK accumulated on the device, one
    read-back; against the CPU restatement with the reference's meter semantics (absent key points = -1 are skipped,
    batch-size weights, ragged last batch)."""
    from oracle.pose_resnet_ref import PoseResNetRef
    from oracle.step_ref import validate_ref
    from uda_poseestimation_amd import synthetic
    from uda_poseestimation_amd.engine import validate
    from uda_poseestimation_amd.lib.models import pose_resnet as pr
    K, S, layers = 5, 128, [1, 1, 1, 1]
    torch.manual_seed(3)
    ref = PoseResNetRef(layers, K)
    torch.nn.init.normal_(ref.head.weight, std=0.05)       # distinct peaks (the default 0.001 head gives near-flat maps)
    # trained-like statistics so that eval mode (running stats) is exercised with non-trivial values
    ref.train()
    with torch.no_grad():
        for i in range(3):
            ref(synthetic.images(4, S, 100 + i))
    net = pr._pose_resnet("t", K, pr.Bottleneck_default, layers, False, False)
    net.load_state_dict(ref.state_dict())
    net = net.cuda()
    net.precision = "fp32"                      # the reference validates in fp32 (no autocast there)
    batches = []
    for i, n in enumerate((4, 4, 3)):
        b = synthetic.mean_teacher_batch(n, num_keypoints=K, image_size=S, heatmap_size=S // 4, seed=20 + i)
        lab, w = b["label_s"].clone(), b["weight_s"].clone()
        if i == 1:
            lab[:, 2] = 0; w[:, 2] = 0          # key point 2 absent from the whole batch -> accuracy -1 -> skipped
        batches.append((b["x_s"], lab, w))
    acc_r, loss_r = validate_ref(batches, ref)
    acc_d, loss_d = validate(batches, net)
    assert len(acc_d) == K
    np.testing.assert_allclose(acc_d, acc_r, atol=1e-6)
    assert abs(loss_d - loss_r) <= 1e-4 * abs(loss_r) + 1e-9
    assert net.training            # validate() restores the mode it found


def test_ema_closed_form_at_full_parameter_count():
    """Size-independent property at the full size (PoseResNet-101: 55.04 M parameters in 325 tensors): with a constant
    student, n EMA steps give teacher_n = a^n * teacher_0 + (1 - a^n) * student, whatever the tensor sizes."""
    import uda_poseestimation_amd.lib.models as models
    from uda_poseestimation_amd import utils as U
    torch.manual_seed(0)
    stu = models.pose_resnet101(num_keypoints=16, pretrained_backbone=False).cuda()
    tea = models.pose_resnet101(num_keypoints=16, pretrained_backbone=False).cuda()
    assert sum(p.numel() for p in stu.parameters()) == 55040568
    t0 = [p.detach().clone() for p in tea.parameters()]
    ema = U.OldWeightEMA(tea, stu, alpha=0.99)
    for p_t, p_s in zip(tea.parameters(), stu.parameters()):
        assert torch.equal(p_t.detach(), p_s.detach())           # the constructor copies student -> teacher (utils.py:14-18)
    with torch.no_grad():
        for p_t, p0 in zip(tea.parameters(), t0):
            p_t.copy_(p0)                                        # start from a teacher that differs from the student
    n = 7
    for _ in range(n):
        ema.step()
    an = 0.99 ** n
    worst = 0.0
    for p_t, p0, p_s in zip(tea.parameters(), t0, stu.parameters()):
        exp = an * p0.double() + (1.0 - an) * p_s.detach().double()
        worst = max(worst, float((p_t.detach().double() - exp).abs().max()))
    assert worst <= 2e-6, worst                                  # n fp32 roundings of O(1) values


def test_heatmap_path_at_benchmark_size_against_oracle():
    """The heat-map kernels at the benchmark's size ([32,16,64,64], and [32,18,96,96] of configs[4]) against the CPU oracle:
    decode, rectify and PCK bit-exact; both losses and their gradients to fp32 rounding; k-th value mask identical."""
    from oracle import keypoints_ref, losses_ref, mean_teacher_ref
    from uda_poseestimation_amd import utils as U
    from uda_poseestimation_amd.lib import keypoint_detection as kd
    from uda_poseestimation_amd.lib.models.loss import ConsLoss, JointsMSELoss
    for (B, K, H, sigma) in ((32, 16, 64, 2), (32, 18, 96, 1.0)):
        g = torch.Generator().manual_seed(K)
        hm = torch.randn(B, K, H, H, generator=g) * 0.3
        hm[:, 0] = -hm[:, 0].abs()                                      # one all-negative key point: decoded as (0, 0)
        hm[1, 1, 5, 7] = hm[1, 1, 20, 3] = 9.0                          # an exact tie: first index wins
        gt = mean_teacher_ref.rectify_ref(torch.randn(B, K, H, H, generator=g), sigma)
        p_ref, v_ref = keypoints_ref.get_max_preds_torch_ref(hm)
        p_dev, v_dev = U.get_max_preds_torch(hm.cuda())
        assert torch.equal(p_dev.cpu(), p_ref) and torch.equal(v_dev.cpu(), v_ref)
        assert torch.equal(U.rectify(hm.cuda(), sigma).cpu(), mean_teacher_ref.rectify_ref(hm, sigma))
        acc_r, avg_r, cnt_r, _ = keypoints_ref.accuracy_ref(hm.numpy(), gt.numpy())
        acc_d, avg_d, cnt_d, _ = kd.accuracy(hm.cuda(), gt.cuda())
        np.testing.assert_allclose(acc_d, acc_r, atol=1e-6)
        assert cnt_d == cnt_r and abs(avg_d - avg_r) < 1e-6
        w = (torch.rand(B, K, 1, generator=g) > 0.2).float()
        p1 = hm.clone().cuda().requires_grad_(True)
        l1 = JointsMSELoss()(p1, gt.cuda(), w.cuda()); l1.backward()
        p2 = hm.clone().requires_grad_(True)
        l2 = losses_ref.joints_mse_ref(p2, gt, w); l2.backward()
        assert abs(float(l1.detach()) - float(l2.detach())) <= 1e-5 * abs(float(l2.detach()))
        np.testing.assert_allclose(p1.grad.cpu().numpy(), p2.grad.numpy(), rtol=1e-5, atol=1e-12)
        mask_r, act_r, thr_r = mean_teacher_ref.conf_mask_ref(hm, 0.5)
        mask_d, _, _ = U.confidence_mask(hm.cuda(), 0.5)
        assert torch.equal(mask_d.cpu().bool(), mask_r.bool())
        s1 = hm.clone().cuda().requires_grad_(True)
        c1 = ConsLoss()(s1, gt.cuda(), tea_mask=mask_d); c1.backward()
        s2 = hm.clone().requires_grad_(True)
        c2 = losses_ref.cons_loss_ref(s2, gt, tea_mask=mask_r); c2.backward()
        assert abs(float(c1.detach()) - float(c2.detach())) <= 1e-5 * abs(float(c2.detach()))
        np.testing.assert_allclose(s1.grad.cpu().numpy(), s2.grad.numpy(), rtol=1e-5, atol=1e-12)


def test_adam_over_all_poseresnet101_parameters_matches_torch():
    """The multi-tensor Adam at the real size (325 tensors, 55.04 M values; the block -> (tensor, offset) table spans them
    all) against torch.optim.Adam on the CPU, three steps."""
    import uda_poseestimation_amd.lib.models as models
    from uda_poseestimation_amd.optim import FusedAdam
    torch.manual_seed(1)
    net = models.pose_resnet101(num_keypoints=16, pretrained_backbone=False).cuda()
    dev_p = list(net.parameters())
    ref_p = [torch.nn.Parameter(p.detach().cpu().contiguous().clone()) for p in dev_p]
    o_ref, o_dev = torch.optim.Adam(ref_p, lr=1e-3), FusedAdam(dev_p, lr=1e-3)
    g = torch.Generator().manual_seed(2)
    for it in range(3):
        for a, b in zip(ref_p, dev_p):
            gr = torch.randn(a.shape, generator=g)
            a.grad = gr
            b.grad = gr.cuda().contiguous(memory_format=torch.channels_last) if gr.dim() == 4 else gr.cuda()
        o_ref.step(); o_dev.step()
    worst = max(float((b.detach().cpu() - a.detach()).abs().max()) for a, b in zip(ref_p, dev_p))
    assert worst <= 2e-6, worst


def test_batchnorm_backward_orthogonality_at_full_size():
    """Size-independent property of the BN backward at a layer1-sized tensor (32 x 64 x 64 pixels, 256 channels; not the
    chunked form) and a layer3-sized one (chunked form): dy = gamma*invstd*(g - mean(g) - xhat*mean(g*xhat)) is orthogonal
    to 1 and to x-hat per channel.  The stored dy is that quantity rounded to bf16 once: its channel sums equal the sums of
    the rounded closed form, and all but a vanishing fraction of its elements are the correctly rounded values."""
    from uda_poseestimation_amd import ops
    for (N, H, C_) in ((32, 64, 256), (32, 16, 1024)):
        g = torch.Generator(device="cuda").manual_seed(C_)
        y = torch.randn(N, H, H, C_, device="cuda", generator=g).bfloat16()
        yf = y.float()
        mean = yf.mean((0, 1, 2)); invstd = 1.0 / torch.sqrt(yf.var((0, 1, 2), unbiased=False) + 1e-5)
        gamma = torch.rand(C_, device="cuda", generator=g) + 0.5
        beta = torch.randn(C_, device="cuda", generator=g) * 0.1
        dz = torch.randn(N, H, H, C_, device="cuda", generator=g)
        dy, dgamma, dbeta, gm = ops.bn_bwd(dz, None, y, gamma, mean, invstd, relu=2, want_g=True, beta=beta)
        M = N * H * H
        xh = ((yf - mean) * invstd).double()
        sc = gamma * invstd; sh = beta - mean * sc
        gd = dz.double() * ((yf * sc + sh) > 0)
        ideal = (gamma * invstd).double() * (gd - gd.sum((0, 1, 2)) / M - xh * (gd * xh).sum((0, 1, 2)) / M)
        assert float(ideal.sum((0, 1, 2)).abs().max()) <= 1e-3 and float((ideal * xh).sum((0, 1, 2)).abs().max()) <= 1e-1   # (fp32 mean / invstd)
        rounded = ideal.float().bfloat16()
        differ = (dy != rounded)
        assert float(differ.float().mean()) <= 2e-3                      # (coefficients are fp32 on the device: rare 1-ulp flips)
        assert float((dy.float() - rounded.float()).abs().max()) <= 2.0 ** -7 * float(ideal.abs().max())
        assert float((dy.double().sum((0, 1, 2)) - rounded.double().sum((0, 1, 2))).abs().max()) <= 5e-2
        np.testing.assert_allclose(dbeta.cpu().numpy(), gd.sum((0, 1, 2)).float().cpu().numpy(), rtol=1e-4, atol=1e-2)
        np.testing.assert_allclose(dgamma.cpu().numpy(), (gd * xh).sum((0, 1, 2)).float().cpu().numpy(), rtol=1e-4, atol=1e-2)
