"""Whole-step property tests on the MI355X: the source-only `pretrain` step of BASELINE.json configs[0]
against the CPU oracle, "loss decreases / PCK does not fall" over >= 50 HIP steps for both step kinds on fixed synthetic
labels, and the state hand-offs between captured (hipGraph) steps and everything else that reads the weights."""
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

def _tiny(K=16, layers=(1, 1, 1, 1), seed=0):
    import uda_poseestimation_amd.lib.models.pose_resnet as pr
    torch.manual_seed(seed)
    return pr._pose_resnet("t", K, pr.Bottleneck_default, list(layers), False, False)


def test_pretrain_step_config0_matches_oracle():
    """BASELINE.json configs[0]: PoseResNet-50 source-only JointsMSELoss step on 16 synthetic 256x256 frames
    (train_human.py:262-289) - engine.pretrain_step on the device vs oracle.step_ref.pretrain_step_ref on the host."""
    import uda_poseestimation_amd.lib.models as models
    from oracle.pose_resnet_ref import pose_resnet50_ref
    from oracle.step_ref import pretrain_step_ref
    from uda_poseestimation_amd import synthetic
    from uda_poseestimation_amd.engine import MeanTeacherTrainer
    torch.manual_seed(0)
    ref = pose_resnet50_ref(16)
    with torch.no_grad():
        for m in ref.modules():
            if hasattr(m, "bn3"):
                m.bn3.weight.fill_(0.25)
    stu = models.pose_resnet50(16, pretrained_backbone=False)
    tea = models.pose_resnet50(16, pretrained_backbone=False)
    stu.load_state_dict(ref.state_dict())
    trainer = MeanTeacherTrainer(stu.cuda(), tea.cuda())
    b = synthetic.mean_teacher_batch(16, seed=4)
    w0 = [p.detach().clone() for p in ref.parameters()]
    out = trainer.pretrain_step(b["x_s"].cuda(), b["label_s"].cuda(), b["weight_s"].cuda())
    opt = torch.optim.Adam(ref.parameters(), lr=1e-4)
    r = pretrain_step_ref(ref, opt, b["x_s"], b["label_s"], b["weight_s"])
    lo, lr_ = float(out["loss_all"]), float(r["loss_all"])
    print(f"configs[0] pretrain step: loss device {lo:.6e} oracle {lr_:.6e}")
    assert abs(lo - lr_) <= 1e-2 * lr_
    agree = total = 0
    for p_dev, p_ref, p0 in zip(stu.parameters(), ref.parameters(), w0):
        d_dev, d_ref = p_dev.detach().cpu() - p0, p_ref.detach() - p0
        sel = d_ref.abs() > 5e-5
        agree += int((torch.sign(d_dev[sel]) == torch.sign(d_ref[sel])).sum())
        total += int(sel.sum())
    print(f"  Adam update sign agreement with the fp32 oracle: {agree / max(total, 1):.4f} over {total} entries")
    assert total > 1e7 and agree / total > 0.85
    # the teacher is untouched by a pretrain step (no EMA there, train_human.py:285-287)
    for a, p0 in zip(tea.parameters(), w0):
        assert torch.equal(a.detach().cpu(), p0)


@pytest.mark.parametrize("kind", ["pretrain", "train"])
def test_loss_decreases_over_150_steps_on_fixed_labels(kind):
    """repeated steps on ONE fixed synthetic batch must drive the supervised loss down and the source PCK
    must not fall (the net memorises the Gaussian labels); for the mean-teacher step the EMA teacher follows the student."""
    from uda_poseestimation_amd import synthetic
    from uda_poseestimation_amd.engine import MeanTeacherTrainer
    from uda_poseestimation_amd.lib import keypoint_detection as kd
    N, K, S = 8, 16, 128
    stu, tea = _tiny(K, seed=1).cuda(), _tiny(K, seed=2).cuda()
    trainer = MeanTeacherTrainer(stu, tea, lr=1e-3, teacher_alpha=0.9, image_size=S, heatmap_size=S // 4)
    b = synthetic.mean_teacher_batch(N, num_keypoints=K, image_size=S, heatmap_size=S // 4, seed=9)
    g = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}
    losses, accs = [], []
    for it in range(150):
        if kind == "pretrain":
            out = trainer.pretrain_step(g["x_s"], g["label_s"], g["weight_s"])
            loss = out["loss_all"]
        else:
            out = trainer.train_step(g["x_s"], g["label_s"], g["weight_s"], g["x_t_stu"], g["x_t_tea"], g["aug_param_stu"], g["aug_param_tea"])
            loss = out["loss_s"]
        losses.append(float(loss))
        accs.append(float(kd.accuracy_device(out["y_s"], g["label_s"])[1][0]))
    first, last = float(np.mean(losses[:5])), float(np.mean(losses[-5:]))
    print(f"{kind}: loss_s {first:.4e} -> {last:.4e}; PCK@0.05 {np.mean(accs[:5]):.3f} -> {np.mean(accs[-5:]):.3f}")
    assert all(np.isfinite(losses)) and last < 0.85 * first
    assert np.mean(accs[-5:]) >= np.mean(accs[:5])
    if kind == "train":
        # EMA with alpha 0.9 over 150 steps: the teacher has left its initial copy and sits between it and the student
        d = sum(float((a.detach() - c.detach()).abs().sum()) for a, c in zip(tea.parameters(), stu.parameters()))
        assert 0 < d < float("inf")


def test_graph_replays_then_eval_forward_at_another_batch_size_sees_fresh_weights():
    """replays of the captured Adam / EMA kernels change parameters without torch's version counters
    moving; an eager forward on ANOTHER executor plan (validate() with another batch size, the teacher after training) must
    not reuse its stale bf16 weight packs."""
    from uda_poseestimation_amd import synthetic
    from uda_poseestimation_amd.engine import GraphedTrainStep, MeanTeacherTrainer
    N, K, S = 4, 16, 128
    stu, tea = _tiny(K, seed=3).cuda(), _tiny(K, seed=3).cuda()
    trainer = MeanTeacherTrainer(stu, tea, lr=1e-2, teacher_alpha=0.5, image_size=S, heatmap_size=S // 4)   # large steps: stale packs would show
    b = synthetic.mean_teacher_batch(N, num_keypoints=K, image_size=S, heatmap_size=S // 4, seed=5)
    g = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}
    args = (g["x_s"], g["label_s"], g["weight_s"], g["x_t_stu"], g["x_t_tea"], g["aug_param_stu"], g["aug_param_tea"])
    xe = synthetic.images(2, S, 77).cuda()                  # another batch size -> another executor plan
    for m in (stu, tea):
        m.eval()
    with torch.no_grad():
        before = [m(xe).clone() for m in (stu, tea)]        # packs of the N=2 plans are now cached
    gs = GraphedTrainStep(trainer, *args, warmup=1)
    for _ in range(6):
        gs.step(*args)
    for m in (stu, tea):
        m.eval()
    with torch.no_grad():
        after = [m(xe).clone() for m in (stu, tea)]
        for m in (stu, tea):                                # force a re-pack of every plan, whatever the caches say
            for hd in m._handles.values():
                hd.wpack_version = None
        forced = [m(xe).clone() for m in (stu, tea)]
    for a, f, b0 in zip(after, forced, before):
        assert torch.equal(a, f), "eval forward after graph replays ran on stale weight packs"
        assert not torch.equal(a, b0)                       # (the weights really moved)
    # ... and the graph itself re-packs inside every replay: a further replay is not disturbed by the eager forwards
    out = gs.step(*args)
    assert torch.isfinite(out["loss_all"])


def test_captured_steps_equal_eager_steps_from_identical_state_with_varying_batches():
    """the split (three-graph, data-parallel form) and unsplit captured steps against an eager twin started from
    IDENTICAL model and optimizer state, over several steps with DIFFERENT batches: parameters and teacher must agree to
    summation-order noise (a wrong mask threshold, a missed gradient sum or a stale pack is orders of magnitude larger)."""
    from uda_poseestimation_amd import synthetic
    from uda_poseestimation_amd.engine import GraphedTrainStep, MeanTeacherTrainer
    N, K, S = 4, 16, 128
    batches = []
    for s in (31, 32, 33, 34):
        b = synthetic.mean_teacher_batch(N, num_keypoints=K, image_size=S, heatmap_size=S // 4, seed=s)
        g = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}
        batches.append((g["x_s"], g["label_s"], g["weight_s"], g["x_t_stu"], g["x_t_tea"], g["aug_param_stu"], g["aug_param_tea"]))
    base = _tiny(K, seed=8)
    for split in (False, True):
        nets = []
        for _ in range(2):
            s_, t_ = _tiny(K, seed=8), _tiny(K, seed=8)
            s_.load_state_dict(base.state_dict())
            nets.append((s_.cuda(), t_.cuda()))
        tr_g = MeanTeacherTrainer(*nets[0], lr=1e-4, image_size=S, heatmap_size=S // 4)
        tr_e = MeanTeacherTrainer(*nets[1], lr=1e-4, image_size=S, heatmap_size=S // 4)
        p0 = [p.detach().clone() for p in nets[0][0].parameters()]
        gs = GraphedTrainStep(tr_g, *batches[0], warmup=1, split=split)       # the warm-up step IS step 1 (on batch 0)
        tr_e.train_step(*batches[0])
        for bt in batches[1:]:
            og = gs.step(*bt)
            oe = tr_e.train_step(*bt)
            assert abs(float(og["loss_all"]) - float(oe["loss_all"])) <= 2e-3 * abs(float(oe["loss_all"]))
            assert abs(float(og["loss_c"]) - float(oe["loss_c"])) <= 5e-3 * abs(float(oe["loss_c"])) + 1e-7
        for (sg, se, tg, te) in ((nets[0][0], nets[1][0], nets[0][1], nets[1][1]),):
            num = den = 0.0
            for pg, pe, q0 in zip(sg.parameters(), se.parameters(), p0):
                num += float(((pg.detach() - pe.detach()) ** 2).sum())
                den += float(((pe.detach() - q0) ** 2).sum())
            rel = (num / max(den, 1e-30)) ** 0.5
            print(f"split={split}: ||student(graph) - student(eager)|| / ||student(eager) - start|| = {rel:.3e} after {len(batches)} steps")
            assert rel < 0.2
            tn = sum(float(((a.detach() - c.detach()) ** 2).sum()) for a, c in zip(tg.parameters(), te.parameters()))
            td = sum(float(((c.detach() - q0) ** 2).sum()) for c, q0 in zip(te.parameters(), p0))
            assert (tn / max(td, 1e-30)) ** 0.5 < 0.1


def test_lr_schedule_reaches_a_captured_step_and_optimizer_checkpoint_round_trip(tmp_path):
    """lr lives in device memory (a MultiStepLR milestone must change what a REPLAYED Adam does),
    replays advance the step counter that state_dict() reports, the optimizer state_dict is torch's plain layout, and a
    checkpoint written by torch.optim.Adam loads."""
    from uda_poseestimation_amd import optim as fo, synthetic
    from uda_poseestimation_amd.engine import GraphedTrainStep, MeanTeacherTrainer
    N, K, S = 4, 16, 128
    stu, tea = _tiny(K, seed=4).cuda(), _tiny(K, seed=4).cuda()
    trainer = MeanTeacherTrainer(stu, tea, lr=1e-3, image_size=S, heatmap_size=S // 4)
    sched = torch.optim.lr_scheduler.MultiStepLR(trainer.stu_optimizer, [1], 0.0)      # lr -> 0 after one scheduler step
    b = synthetic.mean_teacher_batch(N, num_keypoints=K, image_size=S, heatmap_size=S // 4, seed=6)
    g = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}
    args = (g["x_s"], g["label_s"], g["weight_s"], g["x_t_stu"], g["x_t_tea"], g["aug_param_stu"], g["aug_param_tea"])
    gs = GraphedTrainStep(trainer, *args, warmup=1)
    w = [p.detach().clone() for p in stu.parameters()]
    gs.step(*args)
    assert any(not torch.equal(a.detach(), b_) for a, b_ in zip(stu.parameters(), w))          # lr 1e-3: the replay moved the student
    sched.step()
    assert trainer.stu_optimizer.param_groups[0]["lr"] == 0.0
    w = [p.detach().clone() for p in stu.parameters()]
    gs.step(*args)
    assert all(torch.equal(a.detach(), b_) for a, b_ in zip(stu.parameters(), w)), "a replayed Adam ignored the scheduler's lr"
    sd = trainer.stu_optimizer.state_dict()
    assert sd["param_groups"][0]["step"] == 3                                                   # 1 warm-up + 2 replays
    assert {"lr", "betas", "eps", "weight_decay", "grad_scale", "step", "params"} <= set(sd["param_groups"][0])
    assert not any(k.startswith("_") for k in sd["param_groups"][0])                            # no device tables in the checkpoint
    torch.save(sd, tmp_path / "opt.pt")                                                         # no device tables / raw pointers inside
    sd2 = torch.load(tmp_path / "opt.pt", map_location="cpu")
    opt2 = fo.FusedAdam(stu.parameters(), lr=1e-3)
    opt2.load_state_dict(sd2)
    assert opt2.param_groups[0]["step"] == 3
    stu(g["x_s"]).square().mean().backward()
    opt2.step()                                                                                 # CPU-loaded moments move to the device
    assert opt2.state_dict()["param_groups"][0]["step"] == 4
    # a checkpoint written by torch.optim.Adam (per-parameter `step`, no group `step` / `grad_scale`) loads and steps
    ref_opt = torch.optim.Adam(stu.parameters(), lr=1e-3)
    ref_opt.step()
    ref_opt.step()
    opt3 = fo.FusedAdam(stu.parameters(), lr=1e-3)
    opt3.load_state_dict(ref_opt.state_dict())
    assert opt3.param_groups[0]["step"] == 2
    opt3.step()
    assert opt3.state_dict()["param_groups"][0]["step"] == 3


def test_kth_mask_radix_select_matches_torch_for_large_n_and_nan():
    """train_human.py:429-430 at data-parallel sizes (n = 8 ranks x 32 x 21) and beyond; ties; NaN orders as the largest
    value (torch.kthvalue) and a NaN threshold gives an all-false mask."""
    from uda_poseestimation_amd import utils as U
    g = torch.Generator().manual_seed(0)
    for n_rows, K in ((32, 16), (256, 21), (2048, 32)):
        act = torch.rand(n_rows, K, generator=g).cuda()
        act[act < 0.1] = 0.25                                    # many exact ties
        recon = torch.zeros(n_rows, K, 4, 4, device="cuda")
        recon[:, :, 1, 2] = act
        for ratio in (0.5, 0.07, 1.0):
            k = int(ratio * act.numel())
            mask, a, thr = U.confidence_mask(recon, ratio)
            thr_t = torch.kthvalue(act.reshape(-1), k)[0]
            assert torch.equal(a, act) and float(thr) == float(thr_t)
            assert torch.equal(mask, act > thr_t)
    act = torch.rand(8, 16, generator=g).cuda() - 0.5            # negative values too
    act[0, 0] = float("nan"); act[3, 5] = float("nan")
    recon = torch.full((8, 16, 4, 4), -2.0, device="cuda")
    recon[:, :, 0, 0] = act
    a = U.heatmap_activations(recon)
    assert torch.equal(torch.isnan(a), torch.isnan(act))         # amax propagates NaN like torch
    for k_ratio, expect_nan in ((0.5, False), (1.0, True)):
        mask, _, thr = U.confidence_mask(recon, k_ratio, activates=act.clone())
        thr_t = torch.kthvalue(act.reshape(-1), int(k_ratio * act.numel()))[0]
        assert bool(torch.isnan(thr)) == bool(torch.isnan(thr_t)) == expect_nan
        assert torch.equal(mask, act > thr_t)


def test_backward_in_two_parts_equals_whole_backward():
    """udapose_net_backward_part (the cut after layer3's first block that lets a data-parallel step all-reduce 94 % of the
    gradient under the rest of the backward): part 1 leaves exactly the suffix of the flat gradient buffer final, part 2
    completes the prefix, and together they equal the one-call backward (same kernels; weight gradients in two grouped
    launches instead of one).  Then the whole step with the overlap path forced on one process (eager and captured)
    against the plain step."""
    import uda_poseestimation_amd.lib.models.pose_resnet as pr
    from uda_poseestimation_amd import synthetic
    from uda_poseestimation_amd.engine import GraphedTrainStep, MeanTeacherTrainer
    torch.manual_seed(0)
    net = pr._pose_resnet("t", 16, pr.Bottleneck_default, [2, 1, 2, 1], False, False).cuda()
    x = torch.randn(4, 3, 128, 128, generator=torch.Generator().manual_seed(1)).cuda()
    R = torch.randn(4, 16, 32, 32, generator=torch.Generator().manual_seed(2)).cuda()
    net.train()
    (net(x) * R).sum().backward()
    whole = net._flat_grad.clone()
    net.zero_grad(set_to_none=True)
    net._flat_grad.fill_(float("nan"))
    net.split_backward = True
    (net(x) * R).sum().backward()
    net.split_backward = False
    off = net.grad_split_offset()
    names = [n for n, _ in net.named_parameters()]
    first = names[[i for i, p in enumerate(net.parameters()) if sum(q.numel() for q in list(net.parameters())[:i]) == off][0]]
    assert first == "backbone.layer3.0.conv1.weight" and 0 < off < whole.numel() // 2
    upper = net._flat_grad[off:].clone()
    assert torch.isfinite(upper).all() and torch.isnan(net._flat_grad[:off]).all()           # part 1 wrote the suffix and only it
    assert float((upper - whole[off:]).abs().max()) <= 1e-5 * float(whole[off:].abs().max())
    net.finish_backward()
    both = net._flat_grad
    assert torch.isfinite(both).all()
    assert float((both - whole).abs().max()) <= 1e-5 * float(whole.abs().max())
    # ---- whole steps: overlap path forced (no process group: the collectives are skipped, the control flow is the real one)
    N, K, S = 4, 16, 128
    b = synthetic.mean_teacher_batch(N, num_keypoints=K, image_size=S, heatmap_size=S // 4, seed=13)
    g = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}
    args = (g["x_s"], g["label_s"], g["weight_s"], g["x_t_stu"], g["x_t_tea"], g["aug_param_stu"], g["aug_param_tea"])
    base = _tiny(K, seed=5)
    finals = {}
    for mode in ("plain", "overlap_eager", "overlap_graph", "overlap_graph_split"):
        s_, t_ = _tiny(K, seed=5), _tiny(K, seed=5)
        s_.load_state_dict(base.state_dict())
        tr = MeanTeacherTrainer(s_.cuda(), t_.cuda(), image_size=S, heatmap_size=S // 4)
        tr.overlap_allreduce = mode != "plain"
        if "graph" in mode:
            gs = GraphedTrainStep(tr, *args, warmup=1, split=mode.endswith("split"))
            assert gs.overlap and (gs.g_lb2 is not None) == mode.endswith("split")
            outs = [gs.step(*args) for _ in range(2)]
        else:
            outs = [tr.train_step(*args) for _ in range(3)]
        finals[mode] = ([p.detach().clone() for p in s_.parameters()], float(outs[-1]["loss_all"]))
    ref_p, ref_l = finals["plain"]
    p0 = [p.detach().cuda() for p in base.parameters()]
    for mode, (ps, l) in finals.items():
        assert abs(l - ref_l) <= 2e-3 * abs(ref_l), (mode, l, ref_l)
        num = sum(float(((a - b_) ** 2).sum()) for a, b_ in zip(ps, ref_p))
        den = sum(float(((b_ - c) ** 2).sum()) for b_, c in zip(ref_p, p0))
        assert (num / den) ** 0.5 < 0.2, (mode, (num / den) ** 0.5)


def test_fused_optimizer_tail_is_bit_identical_to_adam_then_ema_then_pack():
    """udapose_net_fused_update (Adam + EMA + the weight packs of both plans in one sweep) against the separate launches on
    IDENTICAL gradients and state: parameters, both moments, the teacher and every byte of both pack buffers are equal; a
    second step (moments no longer zero, step counter 2) as well; parameters without gradient (backbone.fc) get the EMA only."""
    from uda_poseestimation_amd import optim as fo
    from uda_poseestimation_amd.utils import OldWeightEMA
    import uda_poseestimation_amd.lib.models.pose_resnet as pr
    x = torch.randn(2, 3, 128, 128, generator=torch.Generator().manual_seed(1)).cuda()
    R = torch.randn(2, 16, 32, 32, generator=torch.Generator().manual_seed(2)).cuda()

    def build():
        torch.manual_seed(3)
        s_ = pr._pose_resnet("t", 16, pr.Bottleneck_default, [2, 1, 1, 1], False, False).cuda()
        torch.manual_seed(4)
        t_ = pr._pose_resnet("t", 16, pr.Bottleneck_default, [2, 1, 1, 1], False, False).cuda()
        opt = fo.FusedAdam(s_.parameters(), lr=1e-3, weight_decay=1e-2)
        ema = OldWeightEMA(t_, s_, alpha=0.9)
        with torch.no_grad():                       # teacher != student, so that the EMA is visible
            for p in t_.parameters():
                p.mul_(1.01)
        return s_, t_, opt, ema

    (sa, ta, oa, ea), (sb, tb, ob, eb) = build(), build()
    for step in range(2):
        # one backward on A; B gets a bit-copy of A's gradients (weight-gradient atomics make two backwards differ in the last bits)
        sa.zero_grad(set_to_none=True); sb.zero_grad(set_to_none=True)
        (sa(x) * R).sum().backward()
        (sb(x) * R).sum().backward()
        with torch.no_grad():
            ta(x); tb(x)
        sb._flat_grad.copy_(sa._flat_grad)
        oa.step(); ea.step()
        assert ob.fused_tail_step(sb, tb, eb) is True
        hd_a, hd_b, ht_a, ht_b = sa._last_hd, sb._last_hd, ta._last_hd, tb._last_hd
        sa.prepare(x)                               # A re-packs from its masters the ordinary way
        with torch.no_grad():
            ta.prepare(x)
        for (n, pa), (_, pb) in zip(sa.named_parameters(), sb.named_parameters()):
            assert torch.equal(pa.detach(), pb.detach()), (step, n)
            if pa.grad is not None:
                assert torch.equal(oa.state[pa]["exp_avg"], ob.state[pb]["exp_avg"]) and torch.equal(oa.state[pa]["exp_avg_sq"], ob.state[pb]["exp_avg_sq"])
        for (n, pa), (_, pb) in zip(ta.named_parameters(), tb.named_parameters()):
            assert torch.equal(pa.detach(), pb.detach()), (step, "teacher", n)
        assert torch.equal(hd_a.wpack, hd_b.wpack), "student packs differ"
        # the teacher plan holds forward packs only: compare through a forward (eval-free: same batch statistics)
        with torch.no_grad():
            assert torch.equal(ta(x), tb(x))
        assert hd_b.wpack_version == (sb.version_key(), True) and ht_b.wpack_version == (tb.version_key(), False)
    assert oa.state_dict()["param_groups"][0]["step"] == ob.state_dict()["param_groups"][0]["step"] == 2
    fc = sb.backbone.fc.weight
    assert fc.grad is None and fc not in ob.state


def test_config2_step_with_style_and_occlusion_captured_equals_eager_twin():
    """BASELINE.json configs[2]'s step (AdaIN style transfer both ways with probability 0.5 each, random alpha;
    adaptive occlusion) captured - style directions as their own graphs with alpha on the device, the occlusion decisions
    taken on the device - against an eager twin with the same host draws from identical state, over steps whose decisions
    differ."""
    from seeded import fill_style_weights
    from uda_poseestimation_amd import synthetic
    from uda_poseestimation_amd.engine import GraphedTrainStep, MeanTeacherTrainer
    from uda_poseestimation_amd.lib.models import Style_net
    fill_style_weights(Style_net.vgg, 11)
    fill_style_weights(Style_net.decoder, 12)
    Style_net.vgg.cuda(); Style_net.decoder.cuda()
    net = Style_net.Net(torch.nn.Sequential(*list(Style_net.vgg.children())[:31]), Style_net.decoder).cuda()
    net.compute_losses = False
    N, K, S = 4, 16, 128
    lo = torch.tensor([-2.1179, -2.0357, -1.8044]).cuda()
    hi = torch.tensor([2.2489, 2.4285, 2.64]).cuda()
    batches = []
    for s in (41, 42, 43, 44, 45, 46):
        b = synthetic.mean_teacher_batch(N, num_keypoints=K, image_size=S, heatmap_size=S // 4, seed=s)
        g = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}
        batches.append((g["x_s"], g["label_s"], g["weight_s"], g["x_t_stu"], g["x_t_tea"], g["aug_param_stu"], g["aug_param_tea"]))
    base = _tiny(K, seed=8)
    nets, trs = [], []
    for _ in range(2):
        s_, t_ = _tiny(K, seed=8), _tiny(K, seed=8)
        s_.load_state_dict(base.state_dict())
        nets.append((s_.cuda(), t_.cuda()))
        tr = MeanTeacherTrainer(*nets[-1], lr=1e-4, image_size=S, heatmap_size=S // 4, style_net=net, recover=(lo, hi), s2t_freq=0.5,
                                t2s_freq=0.5, s2t_alpha=(0.0, 1.0), t2s_alpha=(0.0, 1.0), rng=np.random.RandomState(123),
                                occlude_rate=0.5, occlude_thresh=-1e9, occlude_size=10)     # (threshold below everything: every sample qualifies)
        tr.device_occlusion = True
        trs.append(tr)
    tr_g, tr_e = trs
    p0 = [p.detach().clone() for p in nets[0][0].parameters()]
    with pytest.raises(RuntimeError, match="device_occlusion"):
        tr_g.device_occlusion = False
        GraphedTrainStep(tr_g, *batches[0], warmup=1)
    tr_g.device_occlusion = True
    gs = GraphedTrainStep(tr_g, *batches[0], warmup=1)        # the warm-up step IS step 1 (on batch 0, with step 1's draws)
    tr_e.train_step(*batches[0])
    seen = set()
    probe = np.random.RandomState(123)                         # replay the draws to know which decisions the steps took
    for i in range(len(batches)):
        a = probe.rand() < 0.5
        if a:
            probe.uniform(0, 1)
        b_ = probe.rand() < 0.5
        if b_:
            probe.uniform(0, 1)
        probe.rand(N, 4)
        seen.add((a, b_))
    assert len(seen) >= 3, seen                                # styled / unstyled in both directions do occur
    occl = []
    for bt in batches[1:]:
        og = gs.step(*bt)
        occl.append(int(tr_g.occluded.sum()))
        oe = tr_e.train_step(*bt)
        assert int(tr_e.occluded.sum()) == occl[-1]
        assert torch.isfinite(og["loss_all"])
        assert abs(float(og["loss_all"]) - float(oe["loss_all"])) <= 2e-3 * abs(float(oe["loss_all"]))
        assert abs(float(og["loss_c"]) - float(oe["loss_c"])) <= 5e-3 * abs(float(oe["loss_c"])) + 1e-7
    assert 0 < sum(occl) < N * len(occl)                       # some samples occluded, some not
    num = den = 0.0
    for pg, pe, q0 in zip(nets[0][0].parameters(), nets[1][0].parameters(), p0):
        num += float(((pg.detach() - pe.detach()) ** 2).sum())
        den += float(((pe.detach() - q0) ** 2).sum())
    rel = (num / max(den, 1e-30)) ** 0.5
    print(f"configs[2] captured vs eager twin: relative parameter distance {rel:.3e} after {len(batches)} steps")
    assert rel < 0.2
    assert rng_state_equal(tr_g.rng, tr_e.rng)                 # both consumed exactly the same host draws


def rng_state_equal(a, b):
    sa, sb = a.get_state(), b.get_state()
    return sa[0] == sb[0] and np.array_equal(sa[1], sb[1]) and sa[2:] == sb[2:]


def test_gradient_buffers_summed_inside_the_fused_tail_are_bit_identical_to_the_separate_sum():
    """udapose_net_fused_update(grad2_delta_bytes): the two passes' gradient buffers added inside the Adam / EMA / pack sweep give
    exactly the parameters of axpy-then-sweep, eagerly and captured; p.grad is completed on demand by finish_grads()."""
    from uda_poseestimation_amd import synthetic
    from uda_poseestimation_amd.engine import GraphedTrainStep, MeanTeacherTrainer
    N, K, S = 4, 16, 128
    b = synthetic.mean_teacher_batch(N, num_keypoints=K, image_size=S, heatmap_size=S // 4, seed=6)
    g = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}
    args = (g["x_s"], g["label_s"], g["weight_s"], g["x_t_stu"], g["x_t_tea"], g["aug_param_stu"], g["aug_param_tea"])
    res = {}
    for in_tail in (False, True):
        stu, tea = _tiny(K, seed=4).cuda(), _tiny(K, seed=4).cuda()
        tr = MeanTeacherTrainer(stu, tea, lr=1e-3, image_size=S, heatmap_size=S // 4)
        tr.sum_grads_in_tail = in_tail
        tr.train_step(*args)
        assert tr.fused_last
        assert stu.pending_grad_sum() == 0                     # taken by the tail (or summed before it)
        gs = GraphedTrainStep(tr, *args, warmup=1)
        for _ in range(2):
            gs.step(*args)
        res[in_tail] = [p.detach().clone() for p in list(stu.parameters()) + list(tea.parameters())]
    assert all(torch.equal(a, c) for a, c in zip(res[False], res[True]))
    # a deferred sum is completed by finish_grads() for anyone who wants p.grad itself
    stu, tea = _tiny(K, seed=4).cuda(), _tiny(K, seed=4).cuda()
    tr = MeanTeacherTrainer(stu, tea, image_size=S, heatmap_size=S // 4)
    theta = lambda ap: __import__("uda_poseestimation_amd.warp", fromlist=["x"]).recon_thetas(ap, N, 4.0, "cuda")
    tr._forward_backward(args[0], args[1], args[2], args[3], [args[4]], theta(args[5]), [theta(args[6])])
    torch.cuda.synchronize()
    assert stu.pending_grad_sum() != 0
    part = stu.head.weight.grad.clone()
    stu.finish_grads()
    assert stu.pending_grad_sum() == 0 and not torch.equal(part, stu.head.weight.grad)


def test_deferred_metric_readback_returns_the_synchronous_loops_values():
    """GraphedTrainStep.step_async: losses and device PCK of step i read one step late from a pinned double buffer equal, number for
    number, what a loop that synchronises and reads after every step sees (train_human.py:440-452 logs every iteration); the PCK in
    the metric vector is lib.keypoint_detection.accuracy's."""
    from uda_poseestimation_amd import synthetic
    from uda_poseestimation_amd.engine import GraphedTrainStep, MeanTeacherTrainer
    from uda_poseestimation_amd.lib import keypoint_detection as kd
    N, K, S = 4, 16, 128
    batches = []
    for seed in (3, 4, 5):
        b = synthetic.mean_teacher_batch(N, num_keypoints=K, image_size=S, heatmap_size=S // 4, seed=seed)
        batches.append({k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()})
    arg = lambda g: (g["x_s"], g["label_s"], g["weight_s"], g["x_t_stu"], g["x_t_tea"], g["aug_param_stu"], g["aug_param_tea"])
    # ONE run (round 5): every step's values are read synchronously from the step's own output tensors right after it, and the deferred
    # read-back must deliver exactly those one call later.  (Two separately timed runs - a synchronising loop and a back-to-back one - are
    # not bit-comparable: the stem's weight gradient is accumulated by fp32 atomics in arrival order, and a last-bit difference there
    # passes through Adam's sign-like update into every later loss.  Seen once as a 6 % loss difference in a full-suite run.)
    stu, tea = _tiny(K, seed=9).cuda(), _tiny(K, seed=9).cuda()
    tr = MeanTeacherTrainer(stu, tea, lr=1e-3, image_size=S, heatmap_size=S // 4)
    gs = GraphedTrainStep(tr, *arg(batches[0]), warmup=1)
    sync_vals, deferred = [], []
    for it in range(6):
        g = batches[it % 3]
        m = gs.step_async(*arg(g))
        torch.cuda.synchronize()
        out = gs.out
        _, avg, cnt, _ = kd.accuracy(out["y_s"], g["label_s"])
        sync_vals.append((float(out["loss_all"]), float(out["loss_s"]), float(out["loss_c"]), avg, cnt))
        if it == 0:
            assert m is None
        else:
            deferred.append((m["loss_all"], m["loss_s"], m["loss_c"], m["acc_s"], m["cnt_s"]))
    m = gs.flush_metrics()
    deferred.append((m["loss_all"], m["loss_s"], m["loss_c"], m["acc_s"], m["cnt_s"]))
    assert gs.flush_metrics() is None and len(m["acc_per_keypoint"]) == K
    assert len(sync_vals) == len(deferred) == 6
    for a, c in zip(sync_vals, deferred):
        assert a[0] == c[0] and a[1] == c[1] and a[2] == c[2] and abs(a[3] - c[3]) < 1e-6 and a[4] == c[4], (a, c)


@pytest.mark.parametrize("seed", [123, 4, 14])       # both directions drawn | t2s only | s2t only
def test_config2_eager_step_matches_whole_step_oracle(seed):
    """BASELINE.json configs[2]'s whole step - AdaIN style transfer in both directions (drawn with probability 0.7
    each, random alpha), recover clamp, adaptive occlusion with the reference's host draws - on the device against
    oracle.step_ref.train_step_full_ref (train_human.py:345-438 in the reference's order of np.random draws) from identical
    weights, inputs and generator state, in the reference's precision mix (fp16 student, fp32-grade teacher and style network):
    drawn alphas, stylised inputs, occluded sample set and occluded images, mask, both losses, and the generator state the step
    leaves behind (= the same number and kind of draws were consumed)."""
    from seeded import fill_style_weights
    from oracle.pose_resnet_ref import PoseResNetRef
    from oracle.step_ref import train_step_full_ref
    from oracle.style_ref import make_decoder_ref, make_vgg_ref
    from uda_poseestimation_amd import synthetic
    from uda_poseestimation_amd.engine import MeanTeacherTrainer
    from uda_poseestimation_amd.lib.models import Style_net
    N, K, S, layers = 4, 16, 128, [1, 1, 1, 1]
    fill_style_weights(Style_net.vgg, 11)
    fill_style_weights(Style_net.decoder, 12)
    vgg_ref, dec_ref = make_vgg_ref(), make_decoder_ref()
    vgg_ref.load_state_dict(Style_net.vgg.state_dict())
    dec_ref.load_state_dict(Style_net.decoder.state_dict())
    vgg31_ref = torch.nn.Sequential(*list(vgg_ref.children())[:31]).eval()
    Style_net.vgg.cuda(); Style_net.decoder.cuda()
    net = Style_net.Net(torch.nn.Sequential(*list(Style_net.vgg.children())[:31]), Style_net.decoder).cuda()
    lo_c, hi_c = torch.tensor([-2.1179, -2.0357, -1.8044]), torch.tensor([2.2489, 2.4285, 2.64])
    torch.manual_seed(5)
    ref_s, ref_t = PoseResNetRef(layers, K), PoseResNetRef(layers, K)
    ref_t.load_state_dict(ref_s.state_dict())
    stu, tea = _tiny(K, layers), _tiny(K, layers)
    stu.load_state_dict(ref_s.state_dict())
    b = synthetic.mean_teacher_batch(N, num_keypoints=K, image_size=S, heatmap_size=S // 4, seed=60 + seed)
    g = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}
    # occlusion threshold inside the range of the (random-init) teacher's confidences, so that the candidate sets are data dependent
    ref_t.train()
    with torch.no_grad():
        bufs = {k: v.clone() for k, v in ref_t.state_dict().items() if "running" in k or "num_batches" in k}
        conf0 = ref_t(b["x_t_tea"]).amax(dim=(2, 3))
        ref_t.load_state_dict(bufs, strict=False)
    thresh = float(conf0.flatten().kthvalue(int(0.8 * conf0.numel()))[0])
    kw = dict(s2t_freq=0.7, t2s_freq=0.7, s2t_alpha=(0.2, 1.0), t2s_alpha=(0.2, 1.0), occlude_rate=0.6, occlude_thresh=thresh, occlude_size=10)
    rng_dev, rng_ref = np.random.RandomState(seed), np.random.RandomState(seed)
    tr = MeanTeacherTrainer(stu.cuda(), tea.cuda(), lr=1e-4, image_size=S, heatmap_size=S // 4, style_net=net, recover=(lo_c.cuda(), hi_c.cuda()),
                            rng=rng_dev, precision="reference", **kw)
    tr.device_occlusion = False
    # the effective inputs of the device step, captured on the way
    seen = {}
    orig = tr._forward_backward

    def spy(x_s_in, label_s, weight_s, x_t_stu, x_t_teas_in, theta_stu, thetas_tea):
        seen["x_s_in"], seen["x_t_tea_in"] = x_s_in.detach().clone(), x_t_teas_in[0].detach().clone()
        return orig(x_s_in, label_s, weight_s, x_t_stu, x_t_teas_in, theta_stu, thetas_tea)
    tr._forward_backward = spy
    out = tr.train_step(g["x_s"], g["label_s"], g["weight_s"], g["x_t_stu"], g["x_t_tea"], g["aug_param_stu"], g["aug_param_tea"])
    torch.cuda.synchronize()
    opt = torch.optim.Adam(ref_s.parameters(), lr=1e-4)
    ref = train_step_full_ref(ref_s, ref_t, opt, b["x_s"], b["label_s"], b["weight_s"], b["x_t_stu"], b["x_t_tea"], b["aug_param_stu"],
                              b["aug_param_tea"], ratio=4.0, style=(vgg31_ref, dec_ref), rng=rng_ref, recover=(lo_c, hi_c), image_size=S, **kw)
    # ---- the host generators consumed the same draws (number AND kind: the states are identical)
    st_d, st_r = rng_dev.get_state(), rng_ref.get_state()
    assert st_d[2] == st_r[2] and np.array_equal(st_d[1], st_r[1])
    # ---- style decisions and stylised inputs (fp32-grade style network: 1e-3 of the image range)
    for key, which in (("x_s_in", "alpha_s2t"), ("x_t_tea_in", "alpha_t2s")):
        e = (seen[key].cpu() - ref[key]).abs().max().item()
        print(f"seed {seed}: {which} = {ref[which]}, max|device - oracle| of the effective input {e:.2e}")
        assert e < 5e-4
    # ---- occlusion: same samples, same images up to isolated nearest-neighbour ties of the warps
    assert list(tr.occluded) == list(ref["occluded"]), (tr.occluded, ref["occluded"])
    # ---- mask and losses
    assert torch.equal(out["tea_mask"].cpu().bool(), ref["tea_mask"].bool())                   # the k-th value mask, element for element
    assert abs(float(out["loss_s"]) - float(ref["loss_s"])) <= 1e-3 * float(ref["loss_s"]), (float(out["loss_s"]), float(ref["loss_s"]))
    assert abs(float(out["loss_c"]) - float(ref["loss_c"])) <= 5e-3 * float(ref["loss_c"]) + 1e-7, (float(out["loss_c"]), float(ref["loss_c"]))
    print(f"seed {seed}: occluded {ref['occluded']} (threshold {thresh:.4f}); loss_s {float(out['loss_s']):.6f} / {float(ref['loss_s']):.6f}, "
          f"loss_c {float(out['loss_c']):.4e} / {float(ref['loss_c']):.4e}")
    # EMA and Adam ran: the teacher equals the oracle's teacher to the first update's size
    d = max((a.detach().cpu() - r.detach()).abs().max().item() for a, r in zip(tea.parameters(), ref_t.parameters()))
    assert d < 1e-6 + 1e-3 * 1e-4 * 10, d


def test_fused_sgd_resumes_from_a_torch_sgd_checkpoint_with_its_momentum():
    """torch.optim.SGD checkpoints (train_human.py:136,157,231 save and restore `stu_optimizer`) carry no step
    counter; the restored momentum buffers must NOT be re-initialised by a 'first step'.  torch.optim.SGD (nesterov, weight decay)
    -> state_dict -> FusedSGD -> one step == one further torch step on the same gradients."""
    from uda_poseestimation_amd import optim as fo
    g = torch.Generator().manual_seed(3)
    shapes = [(64, 3, 7, 7), (128,), (33, 17), (256, 64, 1, 1)]
    ps_t = [torch.nn.Parameter(torch.randn(s, generator=g).cuda()) for s in shapes]
    ps_f = [torch.nn.Parameter(p.detach().clone()) for p in ps_t]
    kw = dict(lr=0.05, momentum=0.9, weight_decay=1e-4, nesterov=True)
    ref = torch.optim.SGD(ps_t, **kw)
    for _ in range(3):                                  # three torch steps build up momentum
        for p in ps_t:
            p.grad = torch.randn(p.shape, generator=g).cuda()
        ref.step()
    for a, b in zip(ps_f, ps_t):
        a.data.copy_(b.data)
    import copy
    sd = copy.deepcopy(ref.state_dict())               # (as after torch.save / torch.load: load_state_dict would alias same-device tensors)
    assert "step" not in sd["param_groups"][0]
    opt = fo.FusedSGD(ps_f, **kw)
    opt.load_state_dict(sd)
    assert opt.param_groups[0]["step"] >= 1
    grads = [torch.randn(p.shape, generator=g).cuda() for p in ps_t]
    for a, b, gr in zip(ps_f, ps_t, grads):
        a.grad, b.grad = gr.clone(), gr.clone()
    opt.step()
    ref.step()
    for a, b in zip(ps_f, ps_t):
        assert torch.allclose(a, b, rtol=1e-6, atol=1e-7), (a - b).abs().max()
    for a, b in zip(ps_f, ps_t):                        # the momentum buffers continued, they were not re-seeded with the gradient
        assert torch.allclose(opt.state[a]["momentum_buffer"], ref.state[b]["momentum_buffer"], rtol=1e-6, atol=1e-7)


def test_optimizer_load_state_dict_after_capture_reaches_the_replays():
    """`stu_optimizer.load_state_dict()` between two replays of a captured step.  The restore is IN PLACE (moment
    tensors and device state keep their storage), so the captured Adam continues from the restored moments, step counter and lr -
    checked against an eager twin that loads the same checkpoint; replacing the state tensors wholesale is refused loudly."""
    from uda_poseestimation_amd import synthetic
    from uda_poseestimation_amd.engine import GraphedTrainStep, MeanTeacherTrainer
    N, K, S = 4, 16, 128
    base = _tiny(K, seed=9)
    b = synthetic.mean_teacher_batch(N, num_keypoints=K, image_size=S, heatmap_size=S // 4, seed=12)
    g = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}
    args = (g["x_s"], g["label_s"], g["weight_s"], g["x_t_stu"], g["x_t_tea"], g["aug_param_stu"], g["aug_param_tea"])
    trs = []
    for _ in range(2):
        s_, t_ = _tiny(K, seed=9), _tiny(K, seed=9)
        s_.load_state_dict(base.state_dict())
        trs.append(MeanTeacherTrainer(s_.cuda(), t_.cuda(), lr=1e-3, image_size=S, heatmap_size=S // 4))
    tr_g, tr_e = trs
    gs = GraphedTrainStep(tr_g, *args, warmup=1)
    tr_e.train_step(*args)
    gs.step(*args)
    tr_e.train_step(*args)
    ckpt = {k: (v if not isinstance(v, dict) else v) for k, v in tr_e.stu_optimizer.state_dict().items()}
    import copy
    ckpt = copy.deepcopy(ckpt)
    ckpt["param_groups"][0]["lr"] = 3e-4                              # the checkpoint's lr differs from the running one
    snap_s = copy.deepcopy(tr_e.student.state_dict())
    snap_t = copy.deepcopy(tr_e.teacher.state_dict())
    for tr in (tr_g, tr_e):                                           # "resume": weights and optimizer from the checkpoint
        tr.student.load_state_dict(snap_s)
        tr.teacher.load_state_dict(snap_t)
        ptr0 = tr.stu_optimizer.state[next(iter(tr.student.parameters()))]["exp_avg"].data_ptr()
        tr.stu_optimizer.load_state_dict(copy.deepcopy(ckpt))
        assert tr.stu_optimizer.state[next(iter(tr.student.parameters()))]["exp_avg"].data_ptr() == ptr0      # restored in place
    for _ in range(2):
        og = gs.step(*args)
        oe = tr_e.train_step(*args)
    assert tr_g.stu_optimizer.state_dict()["param_groups"][0]["step"] == tr_e.stu_optimizer.state_dict()["param_groups"][0]["step"] == 4
    num = den = 0.0
    for pg, pe, q in zip(tr_g.student.parameters(), tr_e.student.parameters(), snap_s.values()):
        num += float(((pg.detach() - pe.detach()) ** 2).sum())
    for pe, (k_, q) in zip(tr_e.student.parameters(), [(k_, v) for k_, v in snap_s.items() if "running" not in k_ and "num_batches" not in k_]):
        den += float(((pe.detach() - q.cuda()) ** 2).sum())
    rel = (num / max(den, 1e-30)) ** 0.5
    print(f"replays after an in-place optimizer restore vs the eager twin: relative parameter distance {rel:.3e}")
    assert rel < 1e-3
    # state tensors REPLACED behind the captured launches: refused
    st = tr_g.stu_optimizer.state[next(iter(tr_g.student.parameters()))]
    st["exp_avg"] = st["exp_avg"].clone()
    with pytest.raises(RuntimeError, match="replaced after capture"):
        gs.step(*args)
