"""Parity on a TRAINED-LIKE network: every other whole-network test runs a randomly initialised PoseResNet with
training-mode BatchNorm - the worst case for 16-bit storage (33 bottlenecks each amplify any perturbation).  Here
PoseResNet-101 (reference initialisation) is first trained on the device - fp16 student precision (the reference's autocast
dtype) with the device-side GradScaler, Adam, 400 steps of 8 fresh images each from synthetic.keypoint_batch (images whose
content determines the labels: the network reaches PCK@0.05 ~0.8 on images it has never seen) - then its weights go to the fp32
CPU oracle (oracle/pose_resnet_ref.py) and the bf16 / fp16 / fp32-grade (f16x2) forwards (training-mode BN, N=2, 256x256, unseen
images) and the 16-bit backwards are compared with the oracle: absolute and relative heat-map error, arg-max identity, near-tie rate, and the cosine of
the parameter gradients per stage."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

STAGES = ("stem", "layer1", "layer2", "layer3", "layer4", "upsampling", "head")


def _stage(name):
    if name.startswith("backbone.layer"):
        return name.split(".")[1]
    if name.startswith("backbone."):
        return "stem"
    return name.split(".")[0]


@pytest.fixture(scope="module")
def trained(trained_r101_k16):
    """(device network with trained weights, CPU oracle with the same weights, held-out batches, loss history); the training run is the
    session's (tests/conftest.py::trained_r101_k16)."""
    import uda_poseestimation_amd.lib.models as models
    from oracle.pose_resnet_ref import pose_resnet101_ref
    from uda_poseestimation_amd import synthetic
    sd, hist, _ = trained_r101_k16
    net = models.pose_resnet101(num_keypoints=16, pretrained_backbone=False)
    net.load_state_dict(sd)
    net = net.cuda().train()
    net.precision = "fp16"
    batches = []
    for seed in (5, 6):             # held-out batches (never trained on)
        x, lab, wt = (t.cuda() for t in synthetic.keypoint_batch(8, seed=seed))
        batches.append({"x_s": x, "label_s": lab, "weight_s": wt})
    ref = pose_resnet101_ref(16)
    ref.load_state_dict(sd)
    return net, ref, batches, hist


def test_trained_like_forward_parity_all_precisions(trained):
    """bf16 / fp16 / f16x2 forward vs the fp32 CPU oracle on the trained weights (training-mode BN, N=2, 256x256)."""
    from oracle.keypoints_ref import get_max_preds_ref
    from uda_poseestimation_amd import synthetic
    from uda_poseestimation_amd.lib import keypoint_detection as kd
    net, ref, batches, _ = trained
    x = torch.cat([batches[0]["x_s"][:1].cpu(), batches[1]["x_s"][:1].cpu()])           # two images the network has never seen
    ref.train(); net.train()
    keep = {k: v.clone() for k, v in net.state_dict().items() if "running" in k or "num_batches" in k}
    keep_ref = {k: v.clone() for k, v in ref.state_dict().items() if "running" in k or "num_batches" in k}
    with torch.no_grad():
        y_ref = ref(x)
    ref.load_state_dict(keep_ref, strict=False)
    scale = y_ref.abs().max().item()
    p_ref, _ = get_max_preds_ref(y_ref.numpy())
    fr = y_ref.reshape(32, -1)
    top2 = fr.topk(2, dim=1).values
    rows = {}
    for prec in ("bf16", "fp16", "f16x2"):
        net.precision = prec
        with torch.no_grad():
            y = net(x.cuda())
        net.load_state_dict(keep, strict=False)
        err = (y.cpu() - y_ref).abs().max().item()
        p_dev, _ = kd.get_max_preds(y)
        same = (p_dev.cpu().numpy() == p_ref).all(-1)
        clear = ((top2[:, 0] - top2[:, 1]) > 2 * err).reshape(2, 16).numpy()
        rows[prec] = (err, err / scale, int(same.sum()), 1.0 - clear.mean(), bool(same[clear].all()))
        print(f"trained-like R101 {prec:5s}: max|y|={scale:.3f} max|device - fp32 oracle|={err:.3e} ({err / scale:.2e} of max|y|), arg-max "
              f"identical on {int(same.sum())}/32 key points, near-tie rate (margin <= 2*err) {1.0 - clear.mean():.3f}")
    net.precision = "fp16"
    # the fp32-grade mode meets north_star's absolute bar with identical arg-max
    assert rows["f16x2"][0] < 1e-3 and rows["f16x2"][2] == 32
    # 16-bit storage on a trained-like network (measured: bf16 5.6e-3 = 0.6 % of max|y|, fp16 8.6e-4 - under north_star's 1e-3 -
    # against 18-55 % / 0.6 % on the randomly initialised networks of tests/test_gpu_net.py): arg-max identical wherever the peak
    # margin exceeds the error (and on at least 30 of the 32 key points outright)
    # (Round 6: the network is the same bits in every run - deterministic weight-gradient accumulation - so these are fixed numbers: fp16 1.208e-3,
    # bf16 5.06e-3 = 0.52 % of max|y|, f16x2 4.8e-7.  fp16 sits just ABOVE north_star's absolute 1e-3, and no cheap promotion brings it under: the error is
    # spread evenly over the stages (profiles/r6_attr_fp16.txt: stem 5.0e-4, layer1 5.6e-4, layer2 4.7e-4, layer3 3.6e-4, last deconvolution 4.4e-4,
    # weights alone 4.6e-4; fp32 storage of EVERY activation would still leave 4.9e-4) - DESIGN.md section 4.  Only the fp32-grade f16x2 mode meets 1e-3.)
    assert rows["bf16"][1] < 8e-3 and rows["fp16"][0] < 1.5e-3, rows
    assert rows["fp16"][0] < rows["bf16"][0]
    for prec in ("bf16", "fp16"):
        assert rows[prec][4] and rows[prec][2] >= 30, (prec, rows[prec])


def test_trained_like_gradient_cosine_per_stage(trained):
    """bf16 and fp16 backward (JointsMSE on the trained network's own labels, N=2) vs fp32 autograd of the oracle: cosine and
    relative L2 error of the parameter gradients, per stage."""
    from oracle.losses_ref import joints_mse_ref
    from uda_poseestimation_amd.lib.models.loss import JointsMSELoss
    net, ref, batches, _ = trained
    b = batches[1]
    x, lab, wt = b["x_s"][:2], b["label_s"][:2], b["weight_s"][:2]
    keep = {k: v.clone() for k, v in net.state_dict().items() if "running" in k or "num_batches" in k}
    keep_ref = {k: v.clone() for k, v in ref.state_dict().items() if "running" in k or "num_batches" in k}
    ref.train(); net.train()
    ref.zero_grad()
    joints_mse_ref(ref(x.cpu()), lab.cpu(), wt.cpu()).backward()
    ref.load_state_dict(keep_ref, strict=False)
    g_ref = {n: p.grad.clone() for n, p in ref.named_parameters() if p.grad is not None}
    crit = JointsMSELoss()
    worst = {}
    for prec in ("bf16", "fp16"):
        net.precision = prec
        for p in net.parameters():
            p.grad = None
        (crit(net(x), lab, wt) * 1024.0).backward()          # (scaled so that fp16 gradients stay in range; un-scaled below)
        net.load_state_dict(keep, strict=False)
        acc = {s: [0.0, 0.0, 0.0] for s in STAGES}            # <a,b>, |a|^2, |b|^2 per stage
        for n, p in net.named_parameters():
            if n.startswith("backbone.fc"):
                assert p.grad is None
                continue
            a, r = (p.grad.detach().cpu() / 1024.0).flatten().double(), g_ref[n].flatten().double()
            assert torch.isfinite(a).all(), n
            s = acc[_stage(n)]
            s[0] += float(a @ r); s[1] += float(a @ a); s[2] += float(r @ r)
        line = []
        for st in STAGES:
            ab, aa, rr = acc[st]
            cos = ab / max(np.sqrt(aa * rr), 1e-300)
            rel = np.sqrt(max(aa + rr - 2 * ab, 0.0) / max(rr, 1e-300))
            worst[(prec, st)] = (cos, rel)
            line.append(f"{st} cos {cos:.4f} rel {rel:.2e}")
        print(f"trained-like R101 {prec} parameter gradients vs fp32 autograd: " + " | ".join(line))
    net.precision = "fp16"
    for p in net.parameters():
        p.grad = None
    for (prec, st), (cos, rel) in worst.items():
        assert cos > (0.98 if prec == "fp16" else 0.9), (prec, st, cos, rel)      # measured: fp16 0.992-1.000, bf16 0.939-1.000
