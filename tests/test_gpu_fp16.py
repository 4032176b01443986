"""fp16 element type (the reference's autocast dtype, train_human.py:280,414; BASELINE.json configs[4] "384x384 fp16"):
libudapose_hip_f16.so = the same kernel sources built for elem_t = _Float16 (v_mfma_f32_16x16x32_f16).  Kernel-family parity
through the C ABI against torch-CPU fp32 on the SAME fp16-rounded inputs, the whole network against the oracle (fp16 keeps
10 mantissa bits: tighter than bf16), and the loss-scaled training step with GradScaler semantics kept on the device."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

@pytest.fixture(autouse=True)
def _bf16_unless_stated(monkeypatch):
    """The tests of this module exercise the 16-bit executor: networks start in 'bf16' (BASELINE.json's benched precision) unless a
    test sets another precision.  (A new module's default is 'auto': autocast dtype / fp32-grade teacher, tests/test_gpu_dropin_loop.py.)"""
    from uda_poseestimation_amd.lib.models.pose_resnet import PoseResNet
    monkeypatch.setattr(PoseResNet, "default_precision", "bf16")

H16 = torch.float16


def h(x):
    return x.to(H16).float()


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous().to(H16).cuda()


def nchw(y):
    return y.float().cpu().permute(0, 3, 1, 2).contiguous()


def close(got, ref, tol):
    scale = ref.abs().max().item() + 1e-12
    err = (got - ref).abs().max().item()
    assert err <= tol * scale, f"max err {err:.3e} vs scale {scale:.3e} (tol {tol})"


def test_both_builds_load_and_report_their_element_type():
    from uda_poseestimation_amd import _hip
    assert _hip.lib("bf16").udapose_elem_kind() == 0 and _hip.lib("fp16").udapose_elem_kind() == 1
    assert _hip.lib("bf16") is not _hip.lib("fp16")


CONV_CASES = [
    ("1x1_64_256", 2, 16, 16, 64, 256, 1, 1, 0),
    ("1x1_s2", 2, 16, 16, 256, 512, 1, 2, 0),
    ("3x3_s1", 2, 16, 16, 64, 64, 3, 1, 1),
    ("3x3_s2", 2, 16, 16, 128, 128, 3, 2, 1),
    ("3x3_odd", 1, 9, 7, 64, 64, 3, 2, 1),
    ("1x1_big", 4, 32, 32, 256, 1024, 1, 1, 0),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_fwd_bwd_fp16(case):
    """mirrors tests/test_gpu_kernels.py::test_conv_fwd_bwd; fp16 outputs: 1 ulp = 4.9e-4 relative -> 2e-3 * max"""
    from uda_poseestimation_amd import ops
    _, N, H, W, Ci, Co, K, s, p = case
    g = torch.Generator().manual_seed(1)
    x = h(torch.randn(N, Ci, H, W, generator=g))
    w = h(torch.randn(Co, Ci, K, K, generator=g) / (Ci * K * K) ** 0.5)
    d = ops.conv_desc(N, H, W, Ci, Co, K, s, p)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    ref = F.conv2d(xr, wr, stride=s, padding=p)
    y, stats = ops.conv2d_fwd(nhwc(x), ops.pack_weight(w.cuda(), d, "fwd", dtype=H16), d, want_stats=True)
    assert y.dtype == H16
    close(nchw(y), ref.detach(), 2e-3)
    ssum = stats.double().sum(0).cpu()
    np.testing.assert_allclose(ssum[1].numpy(), (ref.detach().double() ** 2).sum((0, 2, 3)).numpy(), rtol=2e-3)
    dy = h(torch.randn(ref.shape, generator=g))
    ref.backward(dy)
    dx = ops.conv2d_bwd_data(nhwc(dy), ops.pack_weight(w.cuda(), d, "bwd", dtype=H16), d)
    close(nchw(dx), xr.grad, 2e-3)
    dw = ops.conv2d_bwd_weight(nhwc(dy), nhwc(x), d)
    close(dw.cpu().reshape(Co, K, K, Ci).permute(0, 3, 1, 2), wr.grad, 5e-4)


def test_deconv_stem_bn_pool_fp16():
    from uda_poseestimation_amd import ops
    g = torch.Generator().manual_seed(3)
    # ConvTranspose2d 4x4 s2 p1 (Upsampling, pose_resnet.py:33-43)
    N, Hh, Ww, Ci, Co = 2, 8, 8, 256, 256
    x = h(torch.randn(N, Ci, Hh, Ww, generator=g))
    w = h(torch.randn(Ci, Co, 4, 4, generator=g) * 0.05)
    d = ops.conv_desc(N, Hh, Ww, Ci, Co, 4, 2, 1, transposed=True)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    ref = F.conv_transpose2d(xr, wr, stride=2, padding=1)
    y = ops.conv2d_fwd(nhwc(x), ops.pack_weight(w.cuda(), d, "fwd", dtype=H16), d)
    close(nchw(y), ref.detach(), 2e-3)
    dy = h(torch.randn(ref.shape, generator=g))
    ref.backward(dy)
    close(nchw(ops.conv2d_bwd_data(nhwc(dy), ops.pack_weight(w.cuda(), d, "bwd", dtype=H16), d)), xr.grad, 2e-3)
    close(ops.conv2d_bwd_weight(nhwc(dy), nhwc(x), d).cpu().reshape(Ci, 4, 4, Co).permute(0, 3, 1, 2), wr.grad, 5e-4)
    # stem 7x7 s2 on the 3 -> 8 channel padded image
    x = h(torch.randn(2, 3, 64, 64, generator=g))
    w = h(torch.randn(64, 3, 7, 7, generator=g) * 0.1)
    d = ops.conv_desc(2, 64, 64, 8, 64, 7, 2, 3)
    x8 = ops.to_nhwc_bf16(x.cuda(), 8, dtype=H16)
    close(nchw(ops.conv2d_fwd(x8, ops.pack_weight(w.cuda(), d, dtype=H16), d)), F.conv2d(x, w, stride=2, padding=3), 2e-3)
    # training-mode BN apply (+residual, ReLU) and 3x3 s2 max-pool (bit-exact: comparisons only)
    y = h(torch.randn(2, 64, 16, 16, generator=g))
    res = h(torch.randn(2, 64, 16, 16, generator=g))
    bn = torch.nn.BatchNorm2d(64)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(64, generator=g) + 0.5); bn.bias.copy_(torch.randn(64, generator=g) * 0.1)
    zr = F.relu(bn(y) + res)
    stats = torch.stack([y.sum((0, 2, 3)), (y * y).sum((0, 2, 3))])[None].cuda().contiguous()
    z, mean, invstd = ops.bn_train_fwd(nhwc(y), stats, bn.weight.detach().cuda(), bn.bias.detach().cuda(), torch.zeros(64).cuda(),
                                       torch.ones(64).cuda(), torch.zeros((), dtype=torch.long).cuda(), res=nhwc(res), relu=True)
    assert z.dtype == H16
    close(nchw(z), zr.detach(), 2e-3)
    pool, idx = ops.maxpool3x3s2_fwd(nhwc(y))
    assert torch.equal(nchw(pool), F.max_pool2d(y, 3, 2, 1))


def _pair16(layers, K, seed=0, gamma3=0.25):
    import uda_poseestimation_amd.lib.models.pose_resnet as pr
    from oracle.pose_resnet_ref import PoseResNetRef
    torch.manual_seed(seed)
    ref = PoseResNetRef(list(layers), K)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for m in ref.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) * 0.5 + 0.75)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
            if isinstance(m, torch.nn.ConvTranspose2d):
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * (2.0 / (m.weight.shape[0] * 4)) ** 0.5)
            if hasattr(m, "bn3"):
                m.bn3.weight.fill_(gamma3)
        ref.head.weight.copy_(torch.randn(ref.head.weight.shape, generator=g) * 0.05)
    net = pr._pose_resnet("t", K, pr.Bottleneck_default, list(layers), False, False)
    net.load_state_dict(ref.state_dict())
    net = net.cuda()
    net.precision = "fp16"
    return ref, net


def test_network_forward_backward_fp16_vs_oracle():
    """Whole PoseResNet (one / two bottlenecks per stage) in fp16 against the fp32 CPU oracle: 10 mantissa bits of storage
    instead of bf16's 7 -> heat-maps ~8x closer than the bf16 mode on the same network; gradients (loss scale 16 applied
    and removed around the backward) direction and magnitude."""
    ref, net = _pair16((2, 1, 2, 1), 16)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(3, 3, 160, 160, generator=g)
    ref.train(); net.train()
    y_ref = ref(x)
    y = net(x.cuda())
    scale = y_ref.abs().max().item()
    err = (y.cpu() - y_ref.detach()).abs().max().item()
    net.precision = "bf16"
    with torch.no_grad():
        err_bf16 = (net(x.cuda()).cpu() - y_ref.detach()).abs().max().item()
    net.precision = "fp16"
    print(f"fp16 network: max|y|={scale:.3f} |fp16 device - fp32 oracle|={err:.3e} (bf16 mode on the same net: {err_bf16:.3e})")
    assert err <= 2e-2 * scale and err < 0.25 * err_bf16
    R = torch.randn(y_ref.shape, generator=g)
    S = 16.0
    (y_ref * R).sum().backward()
    (y * (R.cuda() * S)).sum().backward()
    worst = 1.0
    for (name, p_r), (_, p_n) in zip(ref.named_parameters(), net.named_parameters()):
        if name.startswith("backbone.fc"):
            continue
        gn = p_n.grad.cpu() / S
        assert torch.isfinite(gn).all(), name
        cos = torch.nn.functional.cosine_similarity(gn.flatten(), p_r.grad.flatten(), dim=0).item()
        worst = min(worst, cos)
        assert cos > 0.95, (name, cos)
    print("fp16 network: worst gradient cosine vs fp32 autograd", worst)


def test_poseresnet101_fp16_vs_bf16_heatmap_error_on_the_conditioned_benched_network():
    """VERDICT r1 #2(d), measured: the benched network (PoseResNet-101, K=16, 256x256, train-mode BN, trained-like conditioning
    bn3.gamma = 0.1 as in tests/test_gpu_net.py) in fp16 - the reference's own autocast dtype, 1927 img/s on configs[1] - and in
    bf16 against the fp32 CPU oracle.  fp16 is ~7x closer (10 mantissa bits against 7: 1.7e-2 against 1.2e-1 on heat-maps of
    maximum 2.8, i.e. 6e-3 against 4e-2 relative) but NO 16-bit storage meets north_star's 1e-3 through 33 train-mode-BN
    bottlenecks of a randomly initialised network; the fp32 mode does (test_gpu_net.py, 3.5e-5).  Arg-max key points agree
    wherever the fp32 peak margin exceeds the error."""
    from test_gpu_net import _pair
    ref, net = _pair((3, 4, 23, 3), 16, seed=3, gamma3=0.1)
    x = torch.randn(2, 3, 256, 256, generator=torch.Generator().manual_seed(11)).clamp(-2.1, 2.6)
    ref.train(); net.train()
    with torch.no_grad():
        y_ref = ref(x)
        net.precision = "fp16"
        y16 = net(x.cuda()).cpu()
        net.precision = "bf16"
        ybf = net(x.cuda()).cpu()
    scale = y_ref.abs().max().item()
    e16, ebf = (y16 - y_ref).abs().max().item(), (ybf - y_ref).abs().max().item()
    print(f"R101 256x256 bn3.gamma=0.1: max|y|={scale:.4f}  |fp16 - fp32 oracle|={e16:.3e}  |bf16 - fp32 oracle|={ebf:.3e}")
    assert e16 <= 1e-2 * scale, (e16, scale)
    assert e16 < 0.25 * ebf
    fr, f16 = y_ref.reshape(32, -1), y16.reshape(32, -1)
    top2 = fr.topk(2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 2 * e16
    assert torch.equal(fr.argmax(1)[clear], f16.argmax(1)[clear])
    print(f"  fp16 arg-max identical on {int((fr.argmax(1) == f16.argmax(1)).sum())}/32 rows ({int(clear.sum())} with a clear fp32 margin)")


def test_fp16_training_step_with_device_side_loss_scaling():
    """MeanTeacherTrainer(precision='fp16'): GradScaler semantics on the device.  (1) a step matches the fp32 oracle step;
    (2) an overflowing gradient (inf) skips the Adam step - parameters AND step counter untouched, EMA still runs
    (train_human.py:437-440) - and halves the scale; (3) the scale grows after `growth_interval` clean steps; all without a
    host read-back (the same sequence runs from a captured hipGraph)."""
    from oracle.pose_resnet_ref import PoseResNetRef
    from oracle.step_ref import train_step_ref
    from uda_poseestimation_amd import synthetic, warp
    from uda_poseestimation_amd.engine import GraphedTrainStep, MeanTeacherTrainer
    import uda_poseestimation_amd.lib.models.pose_resnet as pr
    layers, K, N, S = [1, 1, 1, 1], 16, 4, 128
    torch.manual_seed(0)
    ref_s, ref_t = PoseResNetRef(layers, K), PoseResNetRef(layers, K)
    stu = pr._pose_resnet("t", K, pr.Bottleneck_default, layers, False, False)
    tea = pr._pose_resnet("t", K, pr.Bottleneck_default, layers, False, False)
    stu.load_state_dict(ref_s.state_dict())
    trainer = MeanTeacherTrainer(stu.cuda(), tea.cuda(), image_size=S, heatmap_size=S // 4, precision="fp16", loss_scale_interval=3)
    assert stu.precision == tea.precision == "fp16"
    ref_t.load_state_dict(ref_s.state_dict())
    b = synthetic.mean_teacher_batch(N, num_keypoints=K, image_size=S, heatmap_size=S // 4, seed=7)
    g = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}
    args = (g["x_s"], g["label_s"], g["weight_s"], g["x_t_stu"], g["x_t_tea"], g["aug_param_stu"], g["aug_param_tea"])
    w0 = [p.detach().clone() for p in ref_s.parameters()]
    out = trainer.train_step(*args)
    opt = torch.optim.Adam(ref_s.parameters(), lr=1e-4)
    ref = train_step_ref(ref_s, ref_t, opt, b["x_s"], b["label_s"], b["weight_s"], b["x_t_stu"], b["x_t_tea"], b["aug_param_stu"],
                         b["aug_param_tea"], ratio=4.0)
    assert abs(float(out["loss_s"]) - float(ref["loss_s"])) <= 5e-3 * float(ref["loss_s"])          # UNscaled losses are reported
    assert abs(float(out["loss_c"]) - float(ref["loss_c"])) <= 2e-2 * float(ref["loss_c"]) + 1e-6
    agree = total = 0
    for p_dev, p_ref, p0 in zip(stu.parameters(), ref_s.parameters(), w0):
        d_dev, d_ref = p_dev.detach().cpu() - p0, p_ref.detach() - p0
        sel = d_ref.abs() > 5e-5
        agree += int((torch.sign(d_dev[sel]) == torch.sign(d_ref[sel])).sum())
        total += int(sel.sum())
    print(f"fp16 step: Adam sign agreement with the fp32 oracle {agree / max(total, 1):.4f}")
    assert agree / max(total, 1) > 0.9
    opt_d = trainer.stu_optimizer
    sd = opt_d.state_dict()["param_groups"][0]
    assert sd["step"] == 1 and sd["loss_scale"] == 65536.0 and sd["growth_tracker"] == 1
    # (3) two more clean steps -> growth_interval = 3 reached: scale doubles
    trainer.train_step(*args)
    trainer.train_step(*args)
    sd = opt_d.state_dict()["param_groups"][0]
    assert sd["step"] == 3 and sd["loss_scale"] == 131072.0 and sd["growth_tracker"] == 0
    # (2) overflow: poison one gradient AFTER backward by running the optimizer on hand-made gradients
    ws = [p.detach().clone() for p in stu.parameters()]
    ts = [p.detach().clone() for p in tea.parameters()]
    trainer.stu_optimizer.zero_grad()
    st = trainer._forward_part(args[0], args[1], args[2], args[3], [args[4]], warp.recon_thetas(args[5], N, 4.0, "cuda"),
                               [warp.recon_thetas(args[6], N, 4.0, "cuda")])
    trainer._loss_backward_part(st, None)
    stu.backbone.bn1.weight.grad[5] = float("inf")
    trainer._update()
    sd = opt_d.state_dict()["param_groups"][0]
    assert sd["step"] == 3 and sd["loss_scale"] == 65536.0 and sd["growth_tracker"] == 0          # skipped, backed off
    for p, w in zip(stu.parameters(), ws):
        assert torch.equal(p.detach(), w)                                                         # the student did not move
    # ... and the EMA still ran (teacher := 0.999 t + 0.001 s; train_human.py:438 is outside the scaler)
    for p_t, p_s, t0 in zip(tea.parameters(), stu.parameters(), ts):
        assert torch.equal(p_t.detach(), t0.mul(0.999).add(p_s.detach() * (1.0 - 0.999)))
    # the next clean step applies again
    trainer.train_step(*args)
    assert opt_d.state_dict()["param_groups"][0]["step"] == 4
    # the same machinery from a captured graph: losses finite, counter advances, scale follows the device-side schedule
    gs = GraphedTrainStep(trainer, *args, warmup=1)
    for _ in range(3):
        o = gs.step(*args)
    assert torch.isfinite(o["loss_all"])
    sd = opt_d.state_dict()["param_groups"][0]
    assert sd["step"] == 8 and sd["loss_scale"] in (65536.0, 131072.0, 262144.0)


def test_bench_config4_shape_fp16_runs():
    """BASELINE.json configs[4] as stated: PoseResNet-101, K=18, 384x384 (heat-maps 96x96), fp16, float sigma; through bench.py
    with a small batch (the full b=32 line is a bench run)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1", "--spinup", "0", "--batch", "4", "--image-size", "384",
           "--keypoints", "18", "--sigma", "1.0", "--dtype", "fp16", "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["dtype"] == "fp16" and "configs[4]" in d["config"]["workload"] and d["value"] > 0 and d["loss"] == d["loss"]


def test_config4_full_size_forward_parity_fp16_student_and_f16x2_teacher_vs_oracle():
    """BASELINE.json configs[4] AT FULL SIZE (only a 4-bottleneck bf16 net at 384x384 and a `value > 0` bench run covered it):
    PoseResNet-101, K = 18, 384x384 (heat-maps 96x96), N = 2, training-mode BN, trained-like conditioning (bn3.gamma = 0.1), against
    oracle/pose_resnet_ref.py in fp32 on the CPU - the student's fp16 precision and the teacher's fp32-grade f16x2: absolute error,
    error / max|y|, arg-max identity on the 96x96 maps, and `rectify(sigma = 1.0)` (the animal pipelines' float sigma,
    train_animal.py) of the device's maps bit-exact with the oracle's rectify of the same maps."""
    import uda_poseestimation_amd.lib.models.pose_resnet as pr
    from oracle.keypoints_ref import get_max_preds_ref
    from oracle.mean_teacher_ref import rectify_ref
    from oracle.pose_resnet_ref import PoseResNetRef
    from uda_poseestimation_amd import utils as mt
    from uda_poseestimation_amd.lib import keypoint_detection as kd
    K, S, N = 18, 384, 2
    torch.manual_seed(21)
    ref = PoseResNetRef([3, 4, 23, 3], K)
    g = torch.Generator().manual_seed(22)
    with torch.no_grad():
        for m in ref.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) * 0.5 + 0.75)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
            if isinstance(m, torch.nn.ConvTranspose2d):
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * (2.0 / (m.weight.shape[0] * 4)) ** 0.5)
        ref.head.weight.copy_(torch.randn(ref.head.weight.shape, generator=g) * 0.05)
        ref.head.bias.copy_(torch.randn(ref.head.bias.shape, generator=g) * 0.1)
        for m in ref.modules():          # (after the BN loop above: .modules() visits a block before its bn3)
            if hasattr(m, "bn3"):
                m.bn3.weight.fill_(0.1)
    net = pr._pose_resnet("c4", K, pr.Bottleneck_default, [3, 4, 23, 3], False, False)
    net.load_state_dict(ref.state_dict())
    net = net.cuda()
    x = torch.randn(N, 3, S, S, generator=g)
    ref.train(); net.train()
    keep = {k: v.clone() for k, v in net.state_dict().items() if "running" in k or "num_batches" in k}
    with torch.no_grad():
        y_ref = ref(x)
    assert tuple(y_ref.shape) == (N, K, 96, 96)
    scale = y_ref.abs().max().item()
    p_ref, _ = get_max_preds_ref(y_ref.numpy())
    top2 = y_ref.reshape(N * K, -1).topk(2, dim=1).values
    rows = {}
    for prec in ("fp16", "f16x2"):
        net.precision = prec
        with torch.no_grad():
            y = net(x.cuda())
        net.load_state_dict(keep, strict=False)
        assert net._last_hd.precision == prec and tuple(y.shape) == (N, K, 96, 96)
        err = (y.cpu() - y_ref).abs().max().item()
        p_dev, _ = kd.get_max_preds(y)
        same = (p_dev.cpu().numpy() == p_ref).all(-1)
        clear = ((top2[:, 0] - top2[:, 1]) > 2 * err).reshape(N, K).numpy()
        rows[prec] = (err, err / scale, int(same.sum()), bool(same[clear].all()), float(1.0 - clear.mean()))
        print(f"configs[4] full size R101 K=18 384x384 N=2 {prec:5s}: max|y|={scale:.3f} max|device - fp32 oracle|={err:.3e} ({err / scale:.2e} of max|y|), "
              f"arg-max identical on {int(same.sum())}/{N * K} key points (near-tie rate {1.0 - clear.mean():.3f})")
        # rectify with the animal pipelines' float sigma, on the device's own maps: bit-exact with the oracle's stamp of the same maps
        r_dev = mt.rectify(y, sigma=1.0).cpu()
        r_ref = rectify_ref(y.cpu(), 1.0)
        assert torch.equal(r_dev, r_ref), prec
        assert int((r_dev.reshape(N * K, -1) > 0).sum(1).max()) <= 49
    # the fp32-grade teacher meets north_star's absolute bar at full size with two orders of margin (measured 1.2e-5), arg-max identical
    assert rows["f16x2"][0] < 1e-4 and rows["f16x2"][2] == N * K, rows
    # the fp16 student on this randomly initialised train-mode-BN network (measured 1.6e-2 = 5e-3 of max|y|; a trained network: 8.6e-4,
    # tests/test_gpu_trained.py): within 1 % of the heat-map scale, arg-max identical wherever the peak margin exceeds twice the error
    assert rows["fp16"][1] < 1e-2 and rows["fp16"][3] and rows["fp16"][2] >= N * K - 2, rows
