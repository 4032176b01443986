"""CPU-side tests (no GPU): the C-ABI library loads and exports every symbol the header declares, the host logic of the
drop-in modules matches the oracle / the reference's goldens, the product refuses CPU tensors and never imports the
oracle, and the data-parallel helpers work under a 2-process gloo group."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from uda_poseestimation_amd import _hip
    if not os.path.exists(_hip.LIB_PATH):
        subprocess.run(["make", "-C", os.path.join(ROOT, "uda_poseestimation_amd", "csrc"), "-j8"], check=True)
    return _hip.lib()


def test_library_exports_every_declared_symbol(lib):
    from uda_poseestimation_amd import _hip
    hdr = open(os.path.join(ROOT, "include", "udapose.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(udapose_\w+)\s*\(", hdr)))
    assert len(declared) >= 45
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/udapose.h but not exported"
    assert set(_hip.EXPORTS) <= set(declared), set(_hip.EXPORTS) - set(declared)
    assert lib.udapose_version() >= 200 and lib.udapose_elem_kind() == 0
    lib16 = _hip.lib("fp16")                      # the fp16 build exports the same surface
    for name in declared:
        assert hasattr(lib16, name), f"{name} missing from libudapose_hip_f16.so"
    assert lib16.udapose_elem_kind() == 1
    assert lib.udapose_multi_chunk() > 0          # a pure host query (no GPU needed)


def test_conv_geometry_queries_run_on_host(lib):
    import ctypes as C
    from uda_poseestimation_amd import ops
    for (H, K, s, p, tr, exp) in [(256, 7, 2, 3, False, 128), (64, 3, 1, 1, False, 64), (64, 3, 2, 1, False, 32), (8, 4, 2, 1, True, 16),
                                  (96, 1, 2, 0, False, 48)]:
        d = ops.conv_desc(2, H, H, 64, 64, K, s, p, transposed=tr)
        assert ops.conv_out_hw(d) == (exp, exp)
    d = ops.conv_desc(2, 32, 32, 64, 64, 3, 1, 1, upsample=True, reflect=True)
    assert ops.conv_out_hw(d) == (64, 64)


def test_product_refuses_cpu_tensors_and_never_imports_oracle():
    import uda_poseestimation_amd.lib.models as models
    from uda_poseestimation_amd.lib.models.loss import JointsMSELoss
    from uda_poseestimation_amd import utils as U
    net = models.pose_resnet50(16, pretrained_backbone=False)
    with pytest.raises(RuntimeError, match="MI355X"):
        net(torch.zeros(1, 3, 64, 64))
    with pytest.raises(RuntimeError, match="MI355X"):
        JointsMSELoss()(torch.zeros(1, 2, 4, 4), torch.zeros(1, 2, 4, 4))
    with pytest.raises(RuntimeError, match="MI355X"):
        U.rectify(torch.zeros(1, 2, 8, 8), 2)
    pkg = os.path.join(ROOT, "uda_poseestimation_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f"{f} imports the oracle"


def test_model_factory_contract_matches_reference_usage():
    import uda_poseestimation_amd.lib.models as models
    from oracle.pose_resnet_ref import pose_resnet50_ref, pose_resnet101_ref
    # train_human.py:506-510 discovers architectures like this
    names = sorted(n for n in models.__dict__ if n.islower() and not n.startswith("__") and callable(models.__dict__[n]))
    assert names == ["pose_resnet101", "pose_resnet50"]
    for fac, ref_fac, nparam, ntens in ((models.pose_resnet50, pose_resnet50_ref, 36048440, 172),):
        net, ref = fac(num_keypoints=16, pretrained_backbone=False), ref_fac(16)
        assert list(net.state_dict().keys()) == list(ref.state_dict().keys())
        assert [tuple(p.shape) for p in net.parameters()] == [tuple(p.shape) for p in ref.parameters()]
        assert sum(p.numel() for p in net.parameters()) == nparam and len(list(net.parameters())) == ntens
        # 4-D weights are channels_last in memory (the executor's GEMM layout), logical shapes unchanged
        for p in net.parameters():
            if p.dim() == 4:
                assert p.stride(1) == 1 or p.shape[1] == 1
        # reference checkpoints carry the DataParallel 'module.' prefix (train_human.py:229-230)
        net.load_state_dict({"module." + k: v for k, v in ref.state_dict().items()})
        for a, b in zip(net.parameters(), ref.parameters()):
            assert torch.equal(a.detach(), b.detach())
        assert [len(g["params"]) if isinstance(g["params"], list) else 1 for g in net.get_parameters(lr=0.1)] is not None
    n101 = models.pose_resnet101(num_keypoints=18, pretrained_backbone=False)
    assert sum(p.numel() for p in n101.parameters()) == 55041082 and len(n101.state_dict()) == 646
    with pytest.raises(NotImplementedError):
        from uda_poseestimation_amd.lib.models.pose_resnet import Upsampling
        Upsampling(64, kernel_sizes=(5, 4, 4))


def test_style_net_containers_match_reference_layout():
    from oracle import style_ref
    from uda_poseestimation_amd.lib.models import Style_net
    assert list(Style_net.vgg.state_dict().keys()) == list(style_ref.make_vgg_ref().state_dict().keys())
    assert list(Style_net.decoder.state_dict().keys()) == list(style_ref.make_decoder_ref().state_dict().keys())
    vgg31 = torch.nn.Sequential(*list(Style_net.vgg.children())[:31])
    net = Style_net.Net(vgg31, Style_net.decoder)
    assert [len(list(getattr(net, f"enc_{i}").children())) for i in range(1, 5)] == [4, 7, 7, 13]
    assert all(not p.requires_grad for n in ("enc_1", "enc_2", "enc_3", "enc_4") for p in getattr(net, n).parameters())
    kinds = [s.kind for s in net._enc.steps]
    assert kinds.count("conv") == 9 and kinds.count("pool") == 3 and net._enc.steps[0].pre1x1 is not None
    assert net._enc_taps == [0, 3, 6, 11]                    # relu1_1, relu2_1, relu3_1, relu4_1
    dk = [(s.kind, s.upsample, s.relu) for s in net._dec.steps]
    assert len(dk) == 9 and [u for _, u, _ in dk] == [False, True, False, False, False, True, False, True, False] and dk[-1][2] is False


def test_warp_matrices_and_labels_match_oracle(golden_dir):
    from oracle.affine_ref import inverse_affine_matrix as ref_m
    from uda_poseestimation_amd import synthetic, warp
    rs = np.random.RandomState(1)
    for _ in range(20):
        a, tx, ty, sc, sx, sy = rs.uniform(-180, 180), rs.uniform(-20, 20), rs.uniform(-20, 20), rs.uniform(0.5, 1.5), rs.uniform(-30, 30), rs.uniform(-10, 10)
        assert warp.inverse_affine_matrix(a, [tx, ty], sc, [sx, sy]) == ref_m(a, [tx, ty], sc, [sx, sy])
    ap = synthetic.aug_params(7, np.random.RandomState(2))
    th = warp.recon_thetas(ap, 7, 4.0)
    angle, (tx, ty), (sx, sy), sc = ap
    for i in range(7):
        exp = [ref_m(0.0, [float(tx[i]) / 4, float(ty[i]) / 4], 1.0, [0.0, 0.0]), ref_m(float(angle[i]), [0.0, 0.0], float(sc[i]), [0.0, 0.0]),
               ref_m(0.0, [0.0, 0.0], 1.0, [float(sx[i]), float(sy[i])])]
        np.testing.assert_array_equal(th[i].numpy(), np.array(exp, dtype=np.float32))
    z = np.load(os.path.join(golden_dir, "decode.npz"))
    for b in range(z["kp"].shape[0]):
        t, w = synthetic.gaussian_labels(z["kp"][b], np.ones((16, 1), np.float32), (64, 64), 2, (256, 256))
        np.testing.assert_array_equal(t, z["labels"][b])
        np.testing.assert_array_equal(w, z["weights"][b])
    batch = synthetic.mean_teacher_batch(3, seed=1)
    assert batch["x_s"].shape == (3, 3, 256, 256) and batch["label_s"].shape == (3, 16, 64, 64) and batch["weight_s"].shape == (3, 16, 1)
    assert batch["aug_param_stu"][1][0].dtype == torch.int64 and batch["aug_param_stu"][0].dtype == torch.float64


def _dist_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from uda_poseestimation_amd.engine import GradSync, gather_activates

    class M:
        pass
    m = M()
    m._flat_grad = torch.full((1000,), float(rank + 1))
    GradSync(m)()                                                  # mean over ranks of the flat gradient buffer
    act = torch.arange(6, dtype=torch.float32).reshape(2, 3) + 100 * rank
    allact = gather_activates(act)                                 # rank-major concatenation -> global k-th value
    k = int(0.5 * allact.numel())
    thr = torch.kthvalue(allact, k)[0].item()
    q.put((rank, float(m._flat_grad[0]), allact.tolist(), thr))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_helpers_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_dist_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, g0, allact, thr in res:
        assert g0 == 1.5                                           # (1 + 2) / 2
        assert allact == [0.0, 1.0, 2.0, 3.0, 4.0, 5.0, 100.0, 101.0, 102.0, 103.0, 104.0, 105.0]
        assert thr == 5.0                                          # k = 6 of 12: identical on both ranks (global statistic)


def test_oracle_step_with_a_one_view_list_equals_the_single_view_form():
    """oracle/step_ref.py (train_human.py:358-372): k teacher views as lists; a one-element list is the k = 1 step, and two identical views
    average to the same maps (mean of equal tensors), so all three give one loss."""
    import torch
    from oracle.pose_resnet_ref import PoseResNetRef
    from oracle.step_ref import train_step_ref
    from uda_poseestimation_amd import synthetic
    b = synthetic.mean_teacher_batch(2, num_keypoints=16, image_size=64, heatmap_size=16, seed=1)
    outs = []
    for views in ("scalar", "one", "two"):
        torch.manual_seed(0)
        s_, t_ = PoseResNetRef([1, 1, 1, 1], 16), PoseResNetRef([1, 1, 1, 1], 16)
        t_.load_state_dict(s_.state_dict())
        opt = torch.optim.Adam(s_.parameters(), lr=1e-4)
        xt = b["x_t_tea"] if views == "scalar" else [b["x_t_tea"]] * (1 if views == "one" else 2)
        ap = b["aug_param_tea"] if views == "scalar" else [b["aug_param_tea"]] * (1 if views == "one" else 2)
        o = train_step_ref(s_, t_, opt, b["x_s"], b["label_s"], b["weight_s"], b["x_t_stu"], xt, b["aug_param_stu"], ap, ratio=4.0)
        outs.append((float(o["loss_s"]), float(o["loss_c"]), o["tea_mask"].clone()))
    assert outs[0][0] == outs[1][0] and outs[0][1] == outs[1][1] and torch.equal(outs[0][2], outs[1][2])
    # (two views: the teacher's train-mode BN sees each view separately, running statistics aside the maps are the same)
    assert abs(outs[2][1] - outs[0][1]) <= 1e-6 * abs(outs[0][1]) + 1e-12 and torch.equal(outs[0][2], outs[2][2])


def test_bench_cpu_share_is_bounded_by_the_affinity_mask():
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    n = m.host_cpu_share()
    assert 1 <= n <= len(os.sched_getaffinity(0))


def test_bench_hang_guard_ends_a_rank_that_waits_for_ever():
    """bench.py's _HangGuard (data-parallel runs): a phase that outlives its bound ends the process with the phase's name on stderr
    (exit code 3; the given code when the result line is already out); a cancelled or disabled guard does nothing."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import importlib.util, sys, time\n"
            f"spec = importlib.util.spec_from_file_location('bench_mod', {os.path.join(root, 'bench.py')!r})\n"
            "m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)\n"
            "g = m._HangGuard(); g.arm(0.2, 'disabled'); time.sleep(0.5)\n"
            "g.enabled = True; g.arm(0.2, 'cancelled'); g.cancel(); time.sleep(0.5)\n"
            "g.arm(0.3, 'the phase that hangs', code=int(sys.argv[1])); time.sleep(30)\n")
    for rc_want in (3, 0):
        p = subprocess.run([sys.executable, "-c", code, str(rc_want)], capture_output=True, text=True, timeout=120)
        assert p.returncode == rc_want, (p.returncode, p.stderr[-400:])
        assert "'the phase that hangs' did not finish" in p.stderr and "disabled" not in p.stderr and "cancelled" not in p.stderr
