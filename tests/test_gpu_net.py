"""End-to-end parity of the PoseResNet executor against the CPU oracle (oracle/pose_resnet_ref.py) with identical
weights and inputs.  Bars (BASELINE.json north_star): heat-maps within 1e-3 of the fp32 CPU path, identical arg-max
key-points (checked on maps with a clear peak); parameter gradients compared with a relative L2 tolerance because the
device path computes in bf16 with fp32 accumulation."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu



@pytest.fixture(autouse=True)
def _bf16_unless_stated(monkeypatch):
    """The tests of this module exercise the 16-bit executor: networks start in 'bf16' (BASELINE.json's benched precision) unless a
    test sets another precision.  (A new module's default is 'auto': autocast dtype / fp32-grade teacher, tests/test_gpu_dropin_loop.py.)"""
    from uda_poseestimation_amd.lib.models.pose_resnet import PoseResNet
    monkeypatch.setattr(PoseResNet, "default_precision", "bf16")

def _pair(layers, K, seed=0, gamma3=None):
    import uda_poseestimation_amd.lib.models.pose_resnet as pr
    from oracle.pose_resnet_ref import PoseResNetRef
    torch.manual_seed(seed)
    ref = PoseResNetRef(list(layers), K)
    # make BN affine and the small-init layers non-trivial so that every gradient path is exercised
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for m in ref.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) * 0.5 + 0.75)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
            if isinstance(m, torch.nn.ConvTranspose2d):
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * (2.0 / (m.weight.shape[0] * 4)) ** 0.5)
        ref.head.weight.copy_(torch.randn(ref.head.weight.shape, generator=g) * 0.05)
        ref.head.bias.copy_(torch.randn(ref.head.bias.shape, generator=g) * 0.1)
        if gamma3 is not None:    # "trained-like" conditioning: small residual branches (cf. zero_init_residual)
            for m in ref.modules():
                if hasattr(m, "bn3"):
                    m.bn3.weight.fill_(gamma3)
    net = pr._pose_resnet("test", K, pr.Bottleneck_default, list(layers), False, False)
    net.load_state_dict(ref.state_dict())
    return ref, net.cuda()


def _rel(a, b):
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


def test_state_dict_and_param_order_match_oracle():
    ref, net = _pair((1, 1, 1, 1), 16)
    assert list(ref.state_dict().keys()) == list(net.state_dict().keys())
    for (n1, p1), (n2, p2) in zip(ref.named_parameters(), net.named_parameters()):
        assert n1 == n2 and p1.shape == p2.shape
        assert torch.equal(p1.detach(), p2.detach().cpu())


@pytest.mark.parametrize("layers,N,HW", [((1, 1, 1, 1), 4, 128), ((2, 1, 2, 1), 3, 160)], ids=["tiny128", "small160"])
def test_forward_backward_parity(layers, N, HW):
    from oracle.bf16_emulation import forward_bf16_emulated
    K = 16
    ref, net = _pair(layers, K, gamma3=0.25)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, 3, HW, HW, generator=g)
    ref.train(); net.train()
    with torch.no_grad():
        y_emu = forward_bf16_emulated(ref, x)      # same arithmetic with bf16 storage points: wiring check
    y_ref = ref(x)
    y = net(x.cuda())
    assert y.shape == y_ref.shape and y.dtype == torch.float32
    scale = y_ref.abs().max().item()
    err_emu = (y.cpu() - y_emu).abs().max().item()
    err = (y.cpu() - y_ref.detach()).abs().max().item()
    noise = (y_emu - y_ref.detach()).abs().max().item()   # what bf16 storage alone does to the fp32 result
    print(f"max|y|={scale:.3f} err vs bf16-emulated oracle {err_emu:.3e}, vs fp32 oracle {err:.3e}; emulated-vs-fp32 {noise:.3e}")
    # bars: the device result agrees with the bf16-storage emulation of the SAME arithmetic better than that emulation
    # agrees with fp32 (so what separates device and fp32 oracle is bf16 storage, not the kernels), and the distance to
    # the fp32 oracle is bounded by that storage noise.
    assert err_emu <= 1.0 * noise + 2e-3 * scale, (err_emu, noise, scale)
    assert err <= 2.0 * noise + 2e-3 * scale, (err, noise, scale)
    # BN running statistics follow torch semantics (momentum 0.1, unbiased variance, counter)
    sd_r, sd_n = ref.state_dict(), net.state_dict()
    for k in sd_r:
        if k.endswith("num_batches_tracked"):
            assert int(sd_n[k]) == int(sd_r[k]) == 1
        elif "running" in k:
            np.testing.assert_allclose(sd_n[k].cpu().numpy(), sd_r[k].numpy(), rtol=5e-2, atol=5e-3)
    # ---- backward.  A random linear functional of the heat-maps keeps the BatchNorm backward well conditioned
    # (the MSE gradient lies almost entirely in span{1, xhat} of the last BN layers, whose projection then cancels ~90 %
    # of it and amplifies any forward difference ~10x: that case is covered below with noise-relative bars).
    R = torch.randn(y_ref.shape, generator=g)
    ref.zero_grad()
    (forward_bf16_emulated(ref, x) * R).sum().backward()       # gradients on the same stored (bf16) activations
    (y * R.cuda()).sum().backward()
    report = []
    for (name, p_r), (_, p_n) in zip(ref.named_parameters(), net.named_parameters()):
        if name.startswith("backbone.fc"):
            assert p_n.grad is None and p_r.grad is None     # not part of forward: no gradient, exactly like autograd in the reference
            continue
        r = _rel(p_n.grad.cpu(), p_r.grad)
        cos = torch.nn.functional.cosine_similarity(p_n.grad.cpu().flatten(), p_r.grad.flatten(), dim=0).item()
        report.append((name, r, cos))
    for name, r, cos in report:
        if name.endswith("conv1.weight") or "head" in name or "upsampling" in name:
            print(f"  grad {name:42s} rel {r:.3f} cos {cos:.4f}")
    worst = max(r for _, r, _ in report)
    print("worst relative gradient error (linear functional) vs emulated-storage gradients", worst)
    # Two bf16 evaluations of the same network (device / CPU emulation) differ by ~1 % in their stored activations
    # (accumulation order -> different roundings -> amplified by training-mode BN), and the backward amplifies forward
    # differences further; the tight backward checks are the per-kernel ones (tests/test_gpu_kernels.py: dgrad, wgrad and
    # BN backward against autograd on identical saved tensors).  Here: direction and magnitude of every gradient.
    for name, r, cos in report:
        assert cos > 0.97 and r < 0.30, (name, r, cos)
    # ---- MSE-type loss (JointsMSE shape): compare with fp32 gradients, bounded by what bf16 storage alone does to them
    net.zero_grad(set_to_none=True)
    ref.zero_grad()
    tgt = torch.rand(y_ref.shape, generator=g)
    (0.5 * (ref(x) - tgt) ** 2).mean().backward()
    g_fp32 = {n_: p_.grad.clone() for n_, p_ in ref.named_parameters() if p_.grad is not None}
    ref.zero_grad()
    (0.5 * (forward_bf16_emulated(ref, x) - tgt) ** 2).mean().backward()
    (0.5 * (net(x.cuda()) - tgt.cuda()) ** 2).mean().backward()
    worst_ratio = 0.0
    for (name, p_r), (_, p_n) in zip(ref.named_parameters(), net.named_parameters()):
        if name.startswith("backbone.fc"):
            continue
        noise_g = _rel(p_r.grad, g_fp32[name])                  # emulated-storage vs fp32
        err_g = _rel(p_n.grad.cpu(), g_fp32[name])              # device vs fp32
        worst_ratio = max(worst_ratio, err_g / (noise_g + 0.02))
        assert err_g <= 1.6 * noise_g + 0.03, (name, err_g, noise_g)
    print("MSE loss: worst (device-vs-fp32) / (bf16-storage-vs-fp32) gradient error ratio", worst_ratio)


def test_gradient_accumulation_and_zero_grad():
    ref, net = _pair((1, 1, 1, 1), 16)
    x = torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(9)).cuda()
    net.train()
    net(x).square().mean().backward()
    g1 = [None if p.grad is None else p.grad.clone() for p in net.parameters()]
    assert sum(a is None for a in g1) == 2          # backbone.fc.{weight,bias}: not in forward, no gradient
    # second backward without zero_grad accumulates (two student forwards per step, train_human.py:415-416,436)
    net(x).square().mean().backward()
    for p, a in zip(net.parameters(), g1):
        if a is not None and a.abs().max() > 0:
            assert _rel(p.grad, 2 * a) < 2e-2
    net.zero_grad(set_to_none=True)
    assert all(p.grad is None for p in net.parameters())
    net(x).square().mean().backward()
    for p, a in zip(net.parameters(), g1):
        if a is not None and a.abs().max() > 0:
            assert _rel(p.grad, a) < 2e-2


def test_eval_mode_and_nograd_teacher_forward():
    ref, net = _pair((1, 1, 1, 1), 18)
    x = torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(3))
    ref.eval(); net.eval()
    with torch.no_grad():
        y_ref = ref(x)
        y = net(x.cuda())
    assert (y.cpu() - y_ref).abs().max().item() <= 2.5e-2 * y_ref.abs().max().item()
    # teacher semantics: train() mode under no_grad updates the running stats but keeps no graph
    net.train()
    before = net.backbone.bn1.running_mean.clone()
    with torch.no_grad():
        y2 = net(x.cuda())
    assert not y2.requires_grad and not torch.equal(before, net.backbone.bn1.running_mean)
    # an eval-mode forward WITH grad enabled (a validation loop that forgot no_grad) keeps no backward state and says so at once: one warning, no grad_fn
    net.eval()
    from uda_poseestimation_amd.lib.models.pose_resnet import PoseResNet
    PoseResNet._warned_eval_grad = False
    with pytest.warns(UserWarning, match="eval-mode forward with grad enabled"):
        y3 = net(x.cuda())
    assert y3.grad_fn is None and not y3.requires_grad
    with pytest.raises(RuntimeError):
        y3.sum().backward()


def test_full_size_bf16_heatmaps_bounded_by_storage_noise_argmax_identical_on_clear_peaks():
    """PoseResNet-50, K=16, 256x256, reference init (head / deconv N(0,0.001)), bf16 mode (the benched precision).
    What is asserted: device-vs-fp32 <= 1.5 x (CPU bf16-storage emulation vs fp32) + 1e-4; with trained-like conditioning
    (bn3.gamma = 0.1) additionally < 4e-3 absolute and identical arg-max wherever the peak margin exceeds twice the error.
    north_star's absolute 1e-3 is NOT met in bf16 on a random-init net; the fp32 mode below meets it.
    The tie / near-tie rate north_star asks for is printed."""
    from uda_poseestimation_amd.lib.models import pose_resnet50
    from oracle.pose_resnet_ref import pose_resnet50_ref
    from oracle.bf16_emulation import forward_bf16_emulated
    x = torch.randn(2, 3, 256, 256, generator=torch.Generator().manual_seed(1)).clamp(-2.1, 2.6)
    for gamma3 in (0.1, 1.0):
        torch.manual_seed(0)
        ref = pose_resnet50_ref(16)
        with torch.no_grad():
            for m in ref.modules():
                if hasattr(m, "bn3"):
                    m.bn3.weight.fill_(gamma3)
        net = pose_resnet50(16, pretrained_backbone=False)
        net.load_state_dict(ref.state_dict())
        net = net.cuda()
        ref.train(); net.train()
        with torch.no_grad():
            y_ref = ref(x)
            y_emu = forward_bf16_emulated(ref, x)
            y = net(x.cuda()).cpu()
        assert tuple(y.shape) == (2, 16, 64, 64)
        noise = (y_emu - y_ref).abs().max().item()
        err = (y - y_ref).abs().max().item()
        err_emu = (y - y_emu).abs().max().item()
        print(f"gamma3={gamma3}: max|y|={y_ref.abs().max().item():.4f} |device-fp32|={err:.2e} |device-emulated|={err_emu:.2e} "
              f"|emulated-fp32|={noise:.2e}")
        # bf16 storage alone (CPU emulation) moves the fp32 heat-maps by `noise`; the device must not add to it
        assert err <= 1.5 * noise + 1e-4 and err_emu <= 1.5 * noise + 1e-4
        if gamma3 < 1.0:
            # conditioned like a trained net: absolute bar of north_star scaled by what bf16 storage permits
            assert err < 4e-3
            fr, fy = y_ref.reshape(32, -1), y.reshape(32, -1)
            top2 = fr.topk(2, dim=1).values
            clear = (top2[:, 0] - top2[:, 1]) > 2 * err       # arg-max identity where the peak margin exceeds the error
            assert torch.equal(fr.argmax(1)[clear], fy.argmax(1)[clear])
            same = int((fr.argmax(1) == fy.argmax(1)).sum())
            ties = int((top2[:, 0] == top2[:, 1]).sum())
            print(f"arg-max rows: {same}/32 identical; exact ties {ties}/32, near-ties (margin <= 2*err = {2 * err:.1e}) "
                  f"{32 - int(clear.sum())}/32 (tie / near-tie rate {(32 - int(clear.sum())) / 32:.3f})")


@pytest.mark.parametrize("K,N,HW", [(21, 1, 224), (21, 5, 96), (17, 2, 192), (18, 3, 288)], ids=["k21_n1_224", "k21_n5_96", "k17_n2_192", "k18_n3_288"])
def test_ragged_shapes_key_point_counts_and_batch_sizes(K, N, HW):
    """Shapes off the benchmark's grid: key-point counts that are not multiples of 8 (21: the reference's hand datasets, 17, 18), batch sizes
    1 / 3 / 5, inputs whose layer4 maps are 7x7 / 3x3 / 6x6 / 9x9 (partial tiles in every implicit-GEMM form, odd sub-pixel classes in the
    stride-2 data gradients).  Exact-fp32 forward against the fp32 oracle (indexing faults show as O(1) errors, rounding as 1e-5), then the
    16-bit forward + backward: every gradient's direction and size against the bf16-storage emulation."""
    from oracle.bf16_emulation import forward_bf16_emulated
    ref, net = _pair((1, 1, 1, 1), K, seed=K + N, gamma3=0.25)
    g = torch.Generator().manual_seed(HW)
    x = torch.randn(N, 3, HW, HW, generator=g)
    ref.train(); net.train()
    with torch.no_grad():
        y_ref = ref(x)
        net.precision = "fp32"
        y32 = net(x.cuda())
    assert y32.shape == y_ref.shape == (N, K, HW // 4, HW // 4)
    scale = y_ref.abs().max().item()
    err32 = (y32.cpu() - y_ref).abs().max().item()
    assert err32 <= 2e-4 * scale + 1e-5, (err32, scale)
    net.precision = "bf16"
    net._handles = {}
    y = net(x.cuda())
    with torch.no_grad():
        y_emu = forward_bf16_emulated(ref, x)
    noise = (y_emu - y_ref).abs().max().item()
    assert (y.detach().cpu() - y_ref).abs().max().item() <= 2.0 * noise + 2e-3 * scale
    R = torch.randn(y_ref.shape, generator=g)
    ref.zero_grad()
    (forward_bf16_emulated(ref, x) * R).sum().backward()
    (y * R.cuda()).sum().backward()
    for (name, p_r), (_, p_n) in zip(ref.named_parameters(), net.named_parameters()):
        if name.startswith("backbone.fc"):
            assert p_n.grad is None
            continue
        cos = torch.nn.functional.cosine_similarity(p_n.grad.cpu().flatten(), p_r.grad.flatten(), dim=0).item()
        assert cos > 0.95 and _rel(p_n.grad.cpu(), p_r.grad) < 0.35, (name, cos)


@pytest.mark.parametrize("arch", ["pose_resnet50", "pose_resnet101"])
def test_fp32_mode_meets_the_1e3_heatmap_bar_and_identical_argmax(arch):
    """north_star's bar, on the precision the reference itself uses for the teacher / validate(): exact fp32 MFMA forward
    vs the fp32 CPU oracle, default (reference) initialisation, training-mode BN, full 256x256 input."""
    import uda_poseestimation_amd.lib.models as models
    from oracle import pose_resnet_ref
    from uda_poseestimation_amd.lib import keypoint_detection as kd
    from oracle.keypoints_ref import get_max_preds_ref
    torch.manual_seed(0)
    ref = getattr(pose_resnet_ref, arch + "_ref")(16)
    net = models.__dict__[arch](num_keypoints=16, pretrained_backbone=False)
    net.load_state_dict(ref.state_dict())
    net = net.cuda()
    net.precision = "fp32"
    x = torch.randn(2, 3, 256, 256, generator=torch.Generator().manual_seed(1)).clamp(-2.1, 2.6)
    ref.train(); net.train()
    with torch.no_grad():
        y_ref = ref(x)
        y = net(x.cuda())
    err = (y.cpu() - y_ref).abs().max().item()
    print(f"{arch} fp32 mode: max|y|={y_ref.abs().max().item():.4f} max|device - oracle|={err:.3e}")
    assert err < 1e-3
    p_dev, _ = kd.get_max_preds(y)
    p_ref, _ = get_max_preds_ref(y_ref.numpy())
    top2 = y_ref.reshape(32, -1).topk(2, dim=1).values
    clear = ((top2[:, 0] - top2[:, 1]) > 4 * err).reshape(2, 16).numpy()
    assert np.array_equal(p_dev.cpu().numpy()[clear], p_ref[clear]) and clear.mean() > 0.9
    # running statistics updated exactly like torch
    for k, v in ref.state_dict().items():
        if "running" in k:
            np.testing.assert_allclose(net.state_dict()[k].cpu().numpy(), v.numpy(), rtol=1e-3, atol=1e-4)
    with pytest.raises(RuntimeError):
        net(x.cuda())            # grad-enabled forward is refused in the forward-only precision


def test_grouped_weight_gradients_equal_per_layer_launches():
    """The grouped weight-gradient launch (one launch per tile class over all layers, offset tables, XCD-dealt work units,
    128x128 tiles) against the per-layer launches on the same saved activations: same sums in a different order
    (fp32), every conv weight and BN parameter; also covers accumulation into existing gradients (beta = 1)."""
    from uda_poseestimation_amd import _hip
    ref, net = _pair([2, 2, 2, 2], 6, seed=5)
    lib = _hip.lib()
    x = torch.randn(8, 3, 128, 128, generator=torch.Generator().manual_seed(2)).cuda()
    R = torch.randn(8, 6, 32, 32, generator=torch.Generator().manual_seed(3)).cuda()
    grads = {}
    # mode 1: the grouped launch in its default, deterministic form (round 6: split reductions through per-split partial tiles added in split
    # order, a short split length so that most layers of this small network ARE split); mode 2: the same with fp32 atomics (rounds 1-5);
    # mode 3: mode 1 again - a second, independent plan must give the SAME BITS
    for mode, pol in ((0, {"wgrad_group": 0}), (1, {"wgrad_group": 1, "wgrad_stages": 8}), (2, {"wgrad_group": 1, "wgrad_stages": 8, "wgrad_det": 0}),
                      (3, {"wgrad_group": 1, "wgrad_stages": 8})):
        net.policy, net._handles = pol, {}      # explicit policy of the plans created from here on
        net.zero_grad(set_to_none=True)
        (net(x) * R).sum().backward()
        (net(x) * R).sum().backward()              # second backward accumulates (beta = 1: dst + the ordered sum; atomic adds in mode 2)
        grads[mode] = {n_: p_.grad.clone() for n_, p_ in net.named_parameters() if p_.grad is not None}
    assert len(grads[0]) == len(grads[1]) == len(grads[2]) >= 90
    for n_ in grads[0]:
        for mode in (1, 2):
            a, b = grads[mode][n_], grads[0][n_]
            assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()) + 1e-7, (n_, mode)
        assert torch.equal(grads[1][n_], grads[3][n_]), f"{n_}: two runs of the deterministic backward differ"


@pytest.mark.parametrize("layers,N,HW", [((2, 2, 2, 2), 8, 128), ((1, 2, 1, 1), 3, 160)], ids=["n8_128", "n3_160_odd"])
def test_fused_bn_backward_reduction_equals_separate_reduce_launches(layers, N, HW):
    """Every dgrad launch masks its output with the consumer BatchNorm's ReLU and reduces sum(g), sum(g*xhat) in its epilogue
    (igemm.hip BS mode; net.hip fused chain) against the separate reduce launches on the same saved activations: the same
    bf16 g values summed in a different order (fp32 partials per m-tile instead of per pixel range, fp64 column sums), so
    every parameter gradient agrees to summation-order noise (amplified by bf16 storage of dy further down the chain; the
    bit-level statement is tests/test_gpu_kernels.py::test_dgrad_with_fused_bn_backward_reduction).  Covers the mask-from-z (bn3) and mask-from-y (bn1 / bn2, fp32
    deconv gradients) modes, sub-pixel (stride-2) dgrads with odd sizes, both BN-apply forms and accumulation (beta = 1)."""
    from uda_poseestimation_amd import _hip
    ref, net = _pair(list(layers), 6, seed=7)
    lib = _hip.lib()
    x = torch.randn(N, 3, HW, HW, generator=torch.Generator().manual_seed(2)).cuda()
    R = torch.randn(N, 6, HW // 4, HW // 4, generator=torch.Generator().manual_seed(3)).cuda()
    grads = {}
    for mode, pol in ((0, {"bn_bwd_fused": 0}), (1, {"bn_bwd_fused": 1})):
        net.policy, net._handles = pol, {}
        net.zero_grad(set_to_none=True)
        (net(x) * R).sum().backward()
        (net(x) * R).sum().backward()
        grads[mode] = {n_: p_.grad.clone() for n_, p_ in net.named_parameters() if p_.grad is not None}
    assert len(grads[0]) == len(grads[1]) >= 60
    worst = 0.0
    for n_, mode in [(k, 1) for k in grads[0]]:
        if n_.startswith("backbone.fc"):
            continue
        a, b = grads[mode][n_], grads[0][n_]
        assert torch.isfinite(a).all(), n_
        r = _rel(a, b)
        worst = max(worst, r)
        # the first BN layers of the chain see bit-identical g in both modes: summation-order noise only.  Further down,
        # a 1e-6 change of a BN coefficient flips the bf16 rounding of ~1e-4 of the stored dy elements, which the next layers
        # amplify like any other storage noise (measured profile: 1e-7 at the head, 1e-4 after the deconvs, ~1e-2 at the stem;
        # the same size as the distance between two runs of test_forward_backward_parity's emulation with reordered sums)
        if n_.startswith("upsampling") or n_.startswith("head"):
            assert r < 1e-3, (n_, r)
        else:
            cos = torch.nn.functional.cosine_similarity(a.flatten(), b.flatten(), dim=0).item()
            assert r < 0.12 and cos > 0.99, (n_, r, cos)
    print("fused vs separate BN-backward reduction: worst relative L2 difference of a parameter gradient", worst)


@pytest.mark.parametrize("gamma3", [0.1, 0.25])
def test_poseresnet101_bf16_forward_backward_256_n2_vs_oracle(gamma3):
    """The BENCHED network and precision, whole: PoseResNet-101, K=16, 256x256, bf16, N=2, forward + backward against
    oracle/pose_resnet_ref.py (fp32) and its bf16-storage emulation, with trained-like conditioning (bn3.gamma = 0.1 /
    0.25: 33 train-mode-BN bottlenecks amplify ANY perturbation - with 0.25 the CPU emulation of bf16 storage alone moves
    the fp32 heat-maps by ~18 % of their maximum, so the bars are relative to that storage noise).
    Reports error growth per stage: the fp32 stage outputs' scale, and the device's parameter-gradient error per stage
    group next to what bf16 storage alone does to the same gradients on the CPU."""
    from oracle.bf16_emulation import forward_bf16_emulated
    ref, net = _pair((3, 4, 23, 3), 16, seed=3, gamma3=gamma3)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 3, 256, 256, generator=g).clamp(-2.1, 2.6)
    ref.train(); net.train()
    # ---- forward
    stage_out = {}
    hooks = [m.register_forward_hook(lambda mod, i, o, n_=n_: stage_out.__setitem__(n_, o.detach()))
             for n_, m in (("layer1", ref.backbone.layer1), ("layer2", ref.backbone.layer2), ("layer3", ref.backbone.layer3),
                           ("layer4", ref.backbone.layer4), ("upsampling", ref.upsampling), ("head", ref.head))]
    y_ref = ref(x)
    for h in hooks:
        h.remove()
    y_emu = forward_bf16_emulated(ref, x)
    y = net(x.cuda())
    scale = y_ref.abs().max().item()
    err, err_emu = (y.cpu() - y_ref.detach()).abs().max().item(), (y.cpu() - y_emu.detach()).abs().max().item()
    noise = (y_emu.detach() - y_ref.detach()).abs().max().item()
    print(f"R101 bf16 256x256 N=2 bn3.gamma={gamma3}: max|y|={scale:.4f} |device-fp32|={err:.3e} |device-emulated|={err_emu:.3e} |emulated-fp32|={noise:.3e}")
    for n_, o in stage_out.items():
        print(f"  fp32 stage output {n_:10s} max|.|={o.abs().max().item():.3e} shape {tuple(o.shape)}")
    assert err <= 1.5 * noise + 2e-3 * scale and err_emu <= 1.5 * noise + 2e-3 * scale, (err, err_emu, noise, scale)
    fr, fy = y_ref.detach().reshape(32, -1), y.detach().cpu().reshape(32, -1)
    top2 = fr.topk(2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 2 * err
    assert torch.equal(fr.argmax(1)[clear], fy.argmax(1)[clear])
    print(f"  arg-max identical on {int((fr.argmax(1) == fy.argmax(1)).sum())}/32 rows; near-tie rate {(32 - int(clear.sum())) / 32:.3f}")
    # ---- backward: JointsMSE-shaped loss; gradients vs fp32 autograd, bounded by what bf16 storage alone does to them
    tgt = torch.rand(y_ref.shape, generator=g)
    ref.zero_grad()
    (0.5 * (y_ref - tgt) ** 2).mean().backward()
    g_fp32 = {n_: p_.grad.clone() for n_, p_ in ref.named_parameters() if p_.grad is not None}
    ref.zero_grad()
    (0.5 * (y_emu - tgt) ** 2).mean().backward()
    (0.5 * (y - tgt.cuda()) ** 2).mean().backward()
    groups = {}
    for (name, p_r), (_, p_n) in zip(ref.named_parameters(), net.named_parameters()):
        if name.startswith("backbone.fc"):
            assert p_n.grad is None
            continue
        assert torch.isfinite(p_n.grad).all(), name
        noise_g, err_g = _rel(p_r.grad, g_fp32[name]), _rel(p_n.grad.cpu(), g_fp32[name])
        cos = torch.nn.functional.cosine_similarity(p_n.grad.cpu().flatten(), g_fp32[name].flatten(), dim=0).item()
        key = name.split(".")[1] if name.startswith("backbone.") else name.split(".")[0]
        grp = groups.setdefault(key, [0.0, 0.0, 1.0])
        grp[0], grp[1], grp[2] = max(grp[0], noise_g), max(grp[1], err_g), min(grp[2], cos)
        assert err_g <= 1.6 * noise_g + 0.05, (name, err_g, noise_g)
    for key, (ng, eg, cs) in groups.items():
        print(f"  gradients {key:10s} worst rel err: bf16-storage emulation vs fp32 {ng:.3f} | device vs fp32 {eg:.3f} (min cosine {cs:.4f})")


def test_eval_mode_bn_folded_into_the_conv_epilogue():
    """Policy eval_fold (round 4, the default): in eval mode (validate(), train_human.py:461-500) every convolution's epilogue applies the
    BatchNorm's running-statistics scale / shift, the residual and the ReLU from the fp32 accumulators and writes z itself - against the
    conv -> y -> apply launches (eval_fold = 0): equal to the rounding of the dropped 16-bit y, closer to (never further from) the fp32
    oracle, identical arg-max; in the fp32-grade f16x2 mode both forms sit at the oracle's level."""
    from oracle.pose_resnet_ref import PoseResNetRef
    import uda_poseestimation_amd.lib.models.pose_resnet as pr
    from uda_poseestimation_amd.lib import keypoint_detection as kd
    ref, net = _pair((1, 2, 2, 1), 16, seed=9, gamma3=0.25)
    g = torch.Generator().manual_seed(10)
    with torch.no_grad():
        for m in ref.modules():                      # non-trivial running statistics
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.2)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
    net.load_state_dict(ref.state_dict())
    x = torch.randn(4, 3, 128, 96, generator=g)
    ref.eval(); net.eval()
    with torch.no_grad():
        y_ref = ref(x)
    out = {}
    for prec in ("bf16", "f16x2"):
        for fold in (0, 1):
            net.precision = prec
            net.policy, net._handles = {"eval_fold": fold}, {}
            with torch.no_grad():
                out[(prec, fold)] = net(x.cuda()).cpu()
    scale = y_ref.abs().max().item()
    e = {k: (v - y_ref).abs().max().item() for k, v in out.items()}
    print("eval-mode fold: max|device - fp32 oracle| " + ", ".join(f"{k[0]} fold={k[1]}: {v:.2e}" for k, v in e.items()) + f" (max|y| {scale:.3f})")
    assert (out[("bf16", 0)] - out[("bf16", 1)]).abs().max().item() <= 3e-2 * scale
    assert e[("bf16", 1)] <= 1.2 * e[("bf16", 0)] + 1e-6 and e[("bf16", 1)] <= 2e-2 * scale
    assert e[("f16x2", 1)] < 1e-5 * max(scale, 1.0) and e[("f16x2", 0)] < 1e-5 * max(scale, 1.0)
    p0, _ = kd.get_max_preds(out[("f16x2", 1)].cuda())
    p1, _ = kd.get_max_preds(y_ref.cuda())
    assert torch.equal(p0, p1)


def test_xcd_aligned_batchnorm_kernels_are_bit_identical():
    """Policy bn_xcd_rows (round 4, default 1): the channel-chunked BatchNorm kernels give XCD k all channel chunks of the k-th eighth of the
    pixel rows (the implicit GEMMs' own XCD mapping, so activations cross the kernel boundaries through one L2); 2: the streaming apply
    kernels too.  Only WHICH work-group computes what changes: outputs, running statistics and every gradient are bit for bit those of
    the interleaved order (0), on a network whose layer3 / layer4 take the chunked kernels (>= 1024 pixels, >= 256 channels) and whose
    layer1 / layer2 take the streaming ones."""
    import uda_poseestimation_amd.lib.models.pose_resnet as pr
    torch.manual_seed(11)
    base = pr._pose_resnet("t", 16, pr.Bottleneck_default, [1, 1, 2, 1], False, False)
    x = torch.randn(4, 3, 256, 256, generator=torch.Generator().manual_seed(12)).cuda()
    d = torch.randn(4, 16, 64, 64, generator=torch.Generator().manual_seed(13)).cuda()
    res = {}
    for mode in (0, 1, 2):
        net = pr._pose_resnet("t", 16, pr.Bottleneck_default, [1, 1, 2, 1], False, False)
        net.load_state_dict(base.state_dict())
        net = net.cuda().train()
        # (wgrad_stages = 512: no weight gradient of this size is split over pixels, so none is accumulated by atomics in arrival order -
        # except the stem's, which always is and is compared to rounding below)
        net.policy, net._handles = {"bn_xcd_rows": mode, "wgrad_stages": 512}, {}
        y = net(x)
        y.backward(d)
        torch.cuda.synchronize()
        res[mode] = (y.detach().clone(), {n_: p.grad.clone() for n_, p in net.named_parameters() if p.grad is not None}, [b.clone() for b in net.buffers()])
    for mode in (1, 2):
        assert torch.equal(res[0][0], res[mode][0])
        for n_, g0 in res[0][1].items():
            g1 = res[mode][1][n_]
            if n_ == "backbone.conv1.weight":
                assert (g0 - g1).abs().max().item() <= 1e-5 * g0.abs().max().item() + 1e-12, n_
            else:
                assert torch.equal(g0, g1), (mode, n_)
        assert all(torch.equal(a, b) for a, b in zip(res[0][2], res[mode][2]))


def test_stem_fusion_is_bit_identical_to_separate_launches():
    """Policy stem_fused: BN apply + ReLU + max-pool in one sweep (z of the stem never stored) and the max-pool backward gathered
    inside the BN backward give exactly the outputs, running statistics and gradients of the separate launches - train mode with
    backward (odd tie patterns included: ReLU zeros tie inside pool windows), and eval mode."""
    import uda_poseestimation_amd.lib.models.pose_resnet as pr
    torch.manual_seed(3)
    base = pr._pose_resnet("t", 16, pr.Bottleneck_default, [1, 1, 1, 1], False, False)
    x = torch.randn(3, 3, 160, 96, generator=torch.Generator().manual_seed(5)).cuda()
    d = torch.randn(3, 16, 40, 24, generator=torch.Generator().manual_seed(6)).cuda()
    res = {}
    for fused in (0, 1, 2):       # separate launches | fused forward | fused forward and the pooled backward
        net = pr._pose_resnet("t", 16, pr.Bottleneck_default, [1, 1, 1, 1], False, False)
        net.load_state_dict(base.state_dict())
        net = net.cuda().train()
        net.policy, net._handles = {"stem_fused": fused, "eval_fold": 0}, {}     # (eval_fold: its own test below; here conv -> y -> apply in eval too)
        y = net(x)
        y.backward(d)
        torch.cuda.synchronize()
        grads = [p.grad.clone() for p in net.parameters() if p.grad is not None]
        bufs = [b.clone() for b in net.buffers()]
        net.eval()
        with torch.no_grad():
            ye = net(x)
        res[fused] = (y.detach().clone(), grads, bufs, ye.clone())
    for k in (1, 2):
        assert torch.equal(res[0][0], res[k][0]) and torch.equal(res[0][3], res[k][3])
        assert all(torch.equal(a, b) for a, b in zip(res[0][1], res[k][1]))
        assert all(torch.equal(a, b) for a, b in zip(res[0][2], res[k][2]))
    assert float(res[1][1][0].abs().sum()) > 0          # (the stem's weight gradient is there)


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
def test_relu_bit_mask_of_block_outputs_is_bit_identical_to_reading_z(precision):
    """Policy bn3_mask: every block's BN3 + residual + ReLU apply also saves one bit per element (stored z > 0), and the data gradients
    that mask for that BatchNorm read the byte mask instead of z (1/16 of the bytes).  Same outputs, same gradients, bit for bit -
    including fp16, where small positive values underflow to a stored 0 (the bit follows the STORED value)."""
    import uda_poseestimation_amd.lib.models.pose_resnet as pr
    torch.manual_seed(4)
    base = pr._pose_resnet("t", 16, pr.Bottleneck_default, [2, 1, 2, 1], False, False)
    x = torch.randn(3, 3, 128, 160, generator=torch.Generator().manual_seed(5)).cuda()
    d = torch.randn(3, 16, 32, 40, generator=torch.Generator().manual_seed(6)).cuda() * (1e-4 if precision == "fp16" else 1.0)
    res = {}
    for on in (0, 1):
        net = pr._pose_resnet("t", 16, pr.Bottleneck_default, [2, 1, 2, 1], False, False)
        net.load_state_dict(base.state_dict())
        net = net.cuda().train()
        net.precision = precision
        net.policy, net._handles = {"bn3_mask": on}, {}
        y = net(x)
        y.backward(d)
        torch.cuda.synchronize()
        res[on] = (y.detach().clone(), [p.grad.clone() for p in net.parameters() if p.grad is not None])
    assert torch.equal(res[0][0], res[1][0])
    assert all(torch.equal(a, b) for a, b in zip(res[0][1], res[1][1]))
    assert all(torch.isfinite(g_).all() for g_ in res[1][1]) and float(res[1][1][0].abs().sum()) > 0


def test_deconv_with_bias_matches_oracle():
    """`deconv_with_bias=True` (lib/models/pose_resnet.py:15,41,96 of the reference; never set by its scripts): the three
    ConvTranspose2d carry a bias in front of their BatchNorm.  Forward in training and eval mode (running statistics carry the
    bias) against the fp32 oracle in the fp32-grade mode (1e-4) and in bf16; backward: the bias gradient under a training-mode
    BatchNorm is identically zero (autograd: rounding noise), every other gradient as without the bias."""
    import uda_poseestimation_amd.lib.models.pose_resnet as pr
    from oracle.pose_resnet_ref import PoseResNetRef
    torch.manual_seed(2)
    ref = PoseResNetRef([1, 1, 1, 1], 16, deconv_with_bias=True)
    with torch.no_grad():
        for m in ref.upsampling:
            if isinstance(m, torch.nn.ConvTranspose2d):
                m.bias.normal_(0.0, 0.5)
                m.weight.mul_(30.0)                  # (the reference's std = 0.001 init would make the biases dominate)
    net = pr._pose_resnet("t", 16, pr.Bottleneck_default, [1, 1, 1, 1], False, True)
    assert [k for k in net.state_dict() if k.endswith(".bias") and k.startswith("upsampling.")][:2] == ["upsampling.0.bias", "upsampling.1.bias"]
    net.load_state_dict(ref.state_dict())
    net = net.cuda()
    x = torch.randn(4, 3, 128, 128, generator=torch.Generator().manual_seed(3))
    ref.train(); net.train()
    net.precision = "f16x2"
    with torch.no_grad():
        y_ref = ref(x)
        y = net(x.cuda())
    e = (y.cpu() - y_ref).abs().max().item()
    print(f"deconv_with_bias f16x2 train: max|y| {y_ref.abs().max().item():.3f} err {e:.2e}")
    assert e < 1e-4 * max(1.0, y_ref.abs().max().item())
    for k, v in ref.state_dict().items():
        if "running" in k:
            np.testing.assert_allclose(net.state_dict()[k].cpu().numpy(), v.numpy(), rtol=1e-3, atol=1e-4)
    ref.eval(); net.eval()
    with torch.no_grad():
        e_eval = (net(x.cuda()).cpu() - ref(x)).abs().max().item()
    assert e_eval < 1e-4 * max(1.0, y_ref.abs().max().item()), e_eval
    # bf16, forward + backward
    ref.train(); net.train()
    net.precision = "bf16"
    tgt = torch.rand(y_ref.shape, generator=torch.Generator().manual_seed(4))
    ref.zero_grad()
    yr = ref(x)
    (0.5 * (yr - tgt) ** 2).mean().backward()
    yd = net(x.cuda())
    (0.5 * (yd - tgt.cuda()) ** 2).mean().backward()
    assert (yd.detach().cpu() - yr.detach()).abs().max().item() < 5e-2 * yr.abs().max().item()
    for (name, p_r), (_, p_n) in zip(ref.named_parameters(), net.named_parameters()):
        if name.startswith("backbone.fc"):
            continue
        if name in ("upsampling.0.bias", "upsampling.3.bias", "upsampling.6.bias"):
            assert p_n.grad is not None and float(p_n.grad.abs().max()) == 0.0
            assert float(p_r.grad.abs().max()) < 1e-6 * max(float(g.grad.abs().max()) for g in ref.upsampling.parameters())
        elif name.startswith("upsampling") or name.startswith("head"):
            cos = torch.nn.functional.cosine_similarity(p_n.grad.cpu().flatten(), p_r.grad.flatten(), dim=0).item()
            assert cos > 0.9, (name, cos)          # (bf16 gradients of a random-init train-mode-BN network: 0.92-1.0, as without the bias)


@pytest.mark.parametrize("precision", ["bf16", "f16x2", "fp32"])
def test_forward_only_plan_is_bit_identical_to_the_arena_plan(precision, monkeypatch):
    """Round 5: no-grad forwards - the teacher (train_human.py:346-372, train-mode BN) and validate() (:461-500,
    eval mode) - run a forward-only plan whose y / z tensors rotate through six scratch buffers laid out by liveness instead of a
    bump-allocated arena.  Same kernels, other addresses: heat-maps, running statistics and the deferred-statistics path must be bit
    for bit those of the arena plan; a plan of this kind is ~10x smaller and refuses a backward."""
    import uda_poseestimation_amd.lib.models.pose_resnet as pr
    outs = {}
    for fo in (False, True):
        monkeypatch.setattr(pr.PoseResNet, "fwd_only_plans", fo)
        _, net = _pair((2, 2, 3, 2), 16, seed=5)
        net.precision = precision
        torch.manual_seed(11)
        x = torch.randn(4, 3, 128, 160, device="cuda")
        with torch.no_grad():
            net.train()
            y_tr = net(x).clone()
            y_tr2 = net(x * 0.5).clone()                # (a second call: the scratch buffers are reused, nothing stale may survive)
            y_df = net.forward_deferred_bn(x).clone()
            net.apply_deferred_bn()
            net.eval()
            y_ev = net(x).clone()
        hd = net._last_hd
        assert hd.fwd_only == fo
        bufs = {k: v.clone() for k, v in net.state_dict().items() if "running" in k or "num_batches" in k}
        outs[fo] = (y_tr, y_tr2, y_df, y_ev, bufs, hd.act_bytes)
    for a, b in zip(outs[False][:4], outs[True][:4]):
        assert torch.equal(a, b)
    for k, v in outs[False][4].items():
        assert torch.equal(v, outs[True][4][k]), k
    assert outs[True][5] * 4 < outs[False][5], (outs[True][5], outs[False][5])


def test_forward_only_plan_refuses_backward():
    import ctypes as C
    from uda_poseestimation_amd import _hip
    L = _hip.lib("bf16")
    h = C.c_void_p()
    arr = (C.c_int * 4)(1, 1, 1, 1)
    _hip.check(L.udapose_net_create(arr, 16, 2, 64, 64, 0x200, C.byref(h)), "net_create")
    try:
        g = (C.c_void_p * 1)(None)
        assert L.udapose_net_bind_grads(h, g) != 0
    finally:
        L.udapose_net_destroy(h)
