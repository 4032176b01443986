#!/usr/bin/env python3
"""Benchmark of the mean-teacher UDA pose-estimation hot path on MI355X (BASELINE.json metric).

One STEP = one pass of the reference training step (train_human.py:326-444, k=1) over one synthetic batch:
student forward+backward on x_s and on x_t_stu (N images each), teacher forward on x_t_tea (N images), heat-map
re-warps, JointsMSE + masked consistency loss (rectify, k-th value mask), Adam, EMA teacher update - PoseResNet-101,
K=16, 256x256, N=32 per GPU, bf16 compute / fp32 master weights (BASELINE.json configs[1]).  img/s counts N per step.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \
        bench.py --gpus 8 --steps 10 --warmup 3
    python bench.py --gpus 8 --steps 10 --warmup 3      # no launcher: starts its own 8 ranks (fresh child processes, spawned
                                                        # before this process touches the GPU) and relays rank 0's JSON line

Prints ONE JSON line on rank 0 (contract in the task description), including
  "roofline":     MFMA roofline of the dominant kernel family (implicit-GEMM fprop/dgrad), from HIP events recorded on the
                  launch stream around every convolution launch of ONE eager, instrumented step that runs AFTER the timed
                  region (the timed region is K hipGraph replays and nothing else);
  "cpu_baseline": the CPU oracle's (oracle/step_ref.py, plain torch fp32) throughput on this host, bounded sample.
"""
import argparse
import ctypes
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on this driver stack

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0          # MI355X dense bf16 MFMA peak (/opt/skills/guides/MI355X_MICROARCH.md)
FWD_GFLOP_PER_IMAGE = 24.165       # PoseResNet-101, K=16, 256x256 forward


def _pmc_traffic_file():
    """The newest committed PMC summary (profiles/r<N>_pmc_hbm_traffic.txt)."""
    import glob
    fs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic.txt")))
    return fs[-1] if fs else os.path.join(ROOT, "profiles", "r3_pmc_hbm_traffic.txt")


def measured_traffic_per_igemm_launch():
    """HBM bytes per igemm launch from the committed PMC passes (profiles/r3_pmc_hbm_traffic.txt: separate
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of this benchmark; FETCH_SIZE doubled per the gfx950 note)."""
    try:
        import ast
        d = ast.literal_eval(open(_pmc_traffic_file()).read().strip())
        n, f = d["fetch"]["igemm"]
        _, w = d["write"]["igemm"]
        return (2.0 * f + w) * 1024.0 / n
    except Exception:
        return None


def measured_traffic_per_step(eager_steps=9):
    """HBM bytes per step of the three kernel families (igemm, BatchNorm, weight gradients) from the same committed PMC passes
    (9 eager steps in the profiled run: `--eager --steps 3 --warmup 1` = 1 + 3 timed + 3 synchronised-loop + 2 roofline steps): what the whole step moves, next to the per-launch figure of the dominant kernel."""
    try:
        import ast
        d = ast.literal_eval(open(_pmc_traffic_file()).read().strip())
        return {fam: (2.0 * d["fetch"][fam][1] + d["write"][fam][1]) * 1024.0 / eager_steps for fam in ("igemm", "bn", "wgrad")}
    except Exception:
        return None


def host_cpu_share():
    """Cores this process may actually use: the affinity mask, cut by the cgroup's CPU quota when one is set (a GPU box hands each
    lease a share of the host; running more threads than that share only oversubscribes it)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            q, p = open(path).read().split()[:2]
            if q != "max":
                n = min(n, max(1, int(float(q) / float(p) + 0.5)))
        except Exception:
            pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            n = min(n, max(1, int(q / p + 0.5)))
    except Exception:
        pass
    return n


def cpu_baseline(n, arch_layers, seconds_budget=30.0):
    """CPU oracle step (kind 'port'): PoseResNet-101 mean-teacher step on `n` images per domain, fp32, in the CPU's best configuration
    found inside the budget: the thread count is swept on a short probe (one student forward + backward on 2 images) over
    {8, 16, 32, 64, the process's CPU share, all logical CPUs}, then the whole step is timed with the best count at batch `n`
    (default 8: SURVEY.md's own 8-core probe ran 0.9 img/s there; N = 2 on every logical CPU of the box, round 3's setting, read
    0.32 img/s - oversubscribed threads on a tiny batch)."""
    from oracle.losses_ref import joints_mse_ref
    from oracle.pose_resnet_ref import PoseResNetRef
    from oracle.step_ref import train_step_ref
    from uda_poseestimation_amd import synthetic
    t_start = time.time()
    n = n if n > 0 else 8
    torch.manual_seed(0)
    stu, tea = PoseResNetRef(arch_layers, 16), PoseResNetRef(arch_layers, 16)
    tea.load_state_dict(stu.state_dict())
    opt = torch.optim.Adam(stu.parameters(), lr=1e-4)
    b = synthetic.mean_teacher_batch(n, seed=0)
    share, logical = host_cpu_share(), (os.cpu_count() or 1)
    # (candidates up to twice the share: far more threads than cores only oversubscribes, and one such probe can eat the whole budget)
    cands = sorted({c for c in (8, 16, 32, 64, share, logical) if 1 <= c <= min(logical, max(2 * share, 8))})
    probe = {}
    xs, ls, ws = b["x_s"][:2], b["label_s"][:2], b["weight_s"][:2]
    for c in cands:
        torch.set_num_threads(c)
        best = None
        for it in range(2):                     # (first pass: primitive creation / allocator warm-up)
            t0 = time.time()
            stu.zero_grad()
            joints_mse_ref(stu(xs), ls, ws).backward()
            dt = time.time() - t0
            best = dt if best is None or it > 0 else best
        probe[c] = best
        print(f"cpu_baseline probe: {c} threads {best:.2f} s", file=sys.stderr, flush=True)
        if time.time() - t_start > 0.3 * seconds_budget:
            break
    stu.zero_grad()
    threads = min(probe, key=probe.get)
    torch.set_num_threads(threads)
    times = []
    # (the start state and the FIRST step's outputs are kept: measured_parity() runs the device step from the same weights and inputs)
    ctx = {"sd": {k: v.detach().clone() for k, v in stu.state_dict().items()}, "batch": b, "layers": list(arch_layers), "ref": None}
    for it in range(5):
        t0 = time.time()
        r_ = train_step_ref(stu, tea, opt, b["x_s"], b["label_s"], b["weight_s"], b["x_t_stu"], b["x_t_tea"], b["aug_param_stu"], b["aug_param_tea"])
        times.append(time.time() - t0)
        if it == 0:
            ctx["ref"] = {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in r_.items()}
        print(f"cpu_baseline step {it}: {times[-1]:.2f} s", file=sys.stderr, flush=True)
        if time.time() - t_start + times[-1] > seconds_budget:      # (no room for another iteration)
            break
    best = min(times[1:]) if len(times) > 1 else times[0]
    cpu_baseline.parity_ctx = ctx
    return {"value": round(n / best, 3), "unit": "images/sec", "cores": threads, "kind": "port",
            "sample": f"PoseResNet-101 mean-teacher step (student fwd+bwd on 2x{n}, teacher fwd on {n}, losses, Adam, EMA) fp32 "
                      f"oracle/step_ref.py, batch {n}, {threads} threads (CPU share {share}, {logical} logical CPUs; probe s per 2-image fwd+bwd by threads: "
                      + ", ".join(f"{c}: {t:.2f}" for c, t in probe.items()) + f"), {len(times)} iteration(s), best"
                      + (" of the non-first (times fall over the first iterations: " + ", ".join(f"{t:.1f}" for t in times) + " s)" if len(times) > 1 else "") + f": {best:.2f} s/step"}


def trained_parity_context(arch, threads, n=8):
    """The start state of `measured_parity` on a TRAINED-LIKE network: `arch` trained on the device (synthetic.trained_like_state_dict,
    300 steps of 8 images whose content determines the labels: ~15 s), one fp32 oracle step (oracle/step_ref.py) from those weights on
    a held-out batch of `n` images per domain on the host."""
    from oracle.pose_resnet_ref import PoseResNetRef
    from oracle.step_ref import train_step_ref
    from uda_poseestimation_amd import synthetic
    layers = {"pose_resnet101": [3, 4, 23, 3], "pose_resnet50": [3, 4, 6, 3]}[arch]
    sd, hist, pck = synthetic.trained_like_state_dict(16, steps=300, arch=arch)
    b = synthetic.keypoint_mean_teacher_batch(n, seed=40)
    torch.set_num_threads(threads)
    stu, tea = PoseResNetRef(layers, 16), PoseResNetRef(layers, 16)
    stu.load_state_dict(sd); tea.load_state_dict(sd)
    opt = torch.optim.Adam(stu.parameters(), lr=1e-4)
    r_ = train_step_ref(stu, tea, opt, b["x_s"], b["label_s"], b["weight_s"], b["x_t_stu"], b["x_t_tea"], b["aug_param_stu"], b["aug_param_tea"])
    return {"sd": sd, "batch": b, "layers": layers, "ref": {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in r_.items()},
            "what": f"{arch} K=16 TRAINED on the device (300 steps x 8 synthetic key-point images, JointsMSE {hist[0]:.2e} -> {hist[-1]:.2e}, held-out PCK@0.05 "
                    f"{pck:.2f}), train-mode BN, 256x256, held-out batch of {n} per domain"}


def measured_parity(arch, dev, precision, ctx=None):
    """MEASURED parity of this run (VERDICT r5 #1b): the device step in `precision` ('bf16' | 'fp16' | 'reference') against ONE fp32 oracle
    step (oracle/step_ref.py, train_human.py:326-444) from identical weights and inputs - heat-map max-abs error of the student's
    source pass and of the teacher's re-warped maps, arg-max key points, both losses, the k-th-value mask.  ctx None: the cpu_baseline
    leg's first step = the bench's OWN network (reference initialisation, seed 0, batch 8: randomly initialised train-mode BatchNorm, the
    worst case for 16-bit storage); ctx from trained_parity_context: the same on a trained-like network.  The
    benchmarked size (N = 32, captured) is tests/test_gpu_fullsize.py's."""
    from oracle.keypoints_ref import get_max_preds_ref
    from uda_poseestimation_amd.engine import MeanTeacherTrainer
    from uda_poseestimation_amd.lib import keypoint_detection as kd
    import uda_poseestimation_amd.lib.models as models
    what = None
    if ctx is None:
        ctx = getattr(cpu_baseline, "parity_ctx", None)
        what = f"{arch} K=16 at the reference initialisation (seed 0), train-mode BN, 256x256: the cpu_baseline leg's first oracle step"
    if ctx is None or ctx["ref"] is None or ctx["layers"] != {"pose_resnet101": [3, 4, 23, 3], "pose_resnet50": [3, 4, 6, 3]}[arch]:
        return None
    stu = models.__dict__[arch](num_keypoints=16, pretrained_backbone=False)
    tea = models.__dict__[arch](num_keypoints=16, pretrained_backbone=False)
    stu.load_state_dict(ctx["sd"]); tea.load_state_dict(ctx["sd"])
    trainer = MeanTeacherTrainer(stu.to(dev), tea.to(dev), lr=1e-4, teacher_alpha=0.999, lambda_c=1.0, mask_ratio=0.5, sigma=2, precision=precision)
    b, ref = ctx["batch"], ctx["ref"]
    g = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()}
    out = trainer.train_step(g["x_s"], g["label_s"], g["weight_s"], g["x_t_stu"], g["x_t_tea"], g["aug_param_stu"], g["aug_param_tea"])
    torch.cuda.synchronize()
    y_dev, y_ref = out["y_s"].detach().float().cpu(), ref["y_s"]
    p_ref, _ = get_max_preds_ref(y_ref.numpy())
    p_dev = kd.get_max_preds(out["y_s"].detach().float())[0].cpu().numpy()
    same = (p_dev == p_ref).all(-1)
    m_dev, m_ref = out["tea_mask"].cpu().bool(), ref["tea_mask"].bool()
    ls, lsr, lc, lcr = float(out["loss_s"]), float(ref["loss_s"]), float(out["loss_c"]), float(ref["loss_c"])
    res = {"measured": True, "student": stu._last_hd.precision, "teacher": tea._last_hd.precision, "network": ctx.get("what") or what,
           "batch": int(y_ref.shape[0]), "heatmap_max_abs": (y_dev - y_ref).abs().max().item(), "heatmap_scale": y_ref.abs().max().item(),
           "teacher_heatmap_max_abs": (out["y_t_tea_recon"].detach().float().cpu() - ref["y_t_tea_recon"]).abs().max().item(),
           "argmax_identical": f"{int(same.sum())}/{same.size}", "mask_identical": f"{int((m_dev == m_ref).sum())}/{m_ref.numel()}",
           "loss_s_rel": abs(ls - lsr) / abs(lsr), "loss_c_rel": abs(lc - lcr) / max(abs(lcr), 1e-30), "loss_s": ls, "loss_s_oracle": lsr}
    res["heatmap_rel"] = res["heatmap_max_abs"] / max(res["heatmap_scale"], 1e-30)
    del trainer, stu, tea, g
    torch.cuda.empty_cache()
    return {k: (float(f"{v:.4g}") if isinstance(v, float) else v) for k, v in res.items()}


def style_extras(dev, precision):
    """BASELINE.json configs[2]'s additions to the step: the AdaIN style network (seeded random VGG / decoder: the pretrained files are
    not available offline), both transfer directions forced on with alpha 0.5, adaptive occlusion (train_human.py:345-358,374-412)."""
    import numpy as np
    from uda_poseestimation_amd.lib.models import Style_net
    torch.manual_seed(1)
    Style_net.vgg.to(dev); Style_net.decoder.to(dev)
    style = Style_net.Net(torch.nn.Sequential(*list(Style_net.vgg.children())[:31]), Style_net.decoder).to(dev)
    style.precision = "bf16"           # the fast mode unless precision 'reference' (MeanTeacherTrainer sets 'f16x2' then)
    lo = torch.tensor([-2.1179, -2.0357, -1.8044], device=dev)      # (0 - mean) / std and (1 - mean) / std (train_human.py:32-33)
    hi = torch.tensor([2.2489, 2.4285, 2.64], device=dev)
    return dict(style_net=style, recover=(lo, hi), s2t_freq=1.0, t2s_freq=1.0, s2t_alpha=(0.5, 0.5), t2s_alpha=(0.5, 0.5),
                rng=np.random.RandomState(0), occlude_rate=0.5, occlude_thresh=0.9, occlude_size=10)


def other_config_rate(arch, dev, N, K, S, sigma, dtype, precision, steps=20, warmup=3, config2=False):
    """Another configuration of the step, timed as `steps` hipGraph replays between two synchronisations AFTER the headline's timed
    region (never part of `value`): the precisions that meet north_star's 1e-3 heat-map bar on a trained network - fp16 everywhere, and
    the reference's own mix (train_human.py:346-358,414: fp16 student under the loss scaler, fp32-grade teacher) - and the other
    single-GPU configurations of BASELINE.json: configs[2] (config2 = True: + AdaIN style passes and occlusion, train_human.py:345-412)
    and configs[4]'s workload (K = 18, 384x384, float sigma, fp16: train_animal.py:330-483)."""
    from uda_poseestimation_amd import synthetic
    from uda_poseestimation_amd.engine import GraphedTrainStep, MeanTeacherTrainer
    import uda_poseestimation_amd.lib.models as models
    torch.manual_seed(0)
    student = models.__dict__[arch](num_keypoints=K, pretrained_backbone=False).to(dev)
    teacher = models.__dict__[arch](num_keypoints=K, pretrained_backbone=False).to(dev)
    extra = style_extras(dev, precision) if config2 else {}
    trainer = MeanTeacherTrainer(student, teacher, lr=1e-4, teacher_alpha=0.999, lambda_c=1.0, mask_ratio=0.5, sigma=sigma, image_size=S,
                                 heatmap_size=S // 4, precision=(precision or dtype), **extra)
    trainer.device_occlusion = bool(config2)
    b = synthetic.mean_teacher_batch(N, num_keypoints=K, image_size=S, heatmap_size=S // 4, sigma=sigma, seed=0)
    g = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()}
    graphed = GraphedTrainStep(trainer, g["x_s"], g["label_s"], g["weight_s"], g["x_t_stu"], g["x_t_tea"], g["aug_param_stu"], g["aug_param_tea"])
    for _ in range(warmup):
        out = graphed.step(None, None, None, None, None, g["aug_param_stu"], g["aug_param_tea"])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = graphed.step(None, None, None, None, None, g["aug_param_stu"], g["aug_param_tea"])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    loss = float(out["loss_all"])
    res = {"images_per_sec": round(N * steps / dt, 2), "ms_per_step": round(dt / steps * 1e3, 3), "steps": steps, "loss_finite": loss == loss,
           "student": student._last_hd.precision, "teacher": teacher._last_hd.precision}
    if config2:
        sn = extra["style_net"]
        res["style_net"] = getattr(sn, "precision", None)
    del graphed, trainer, student, teacher, g, extra
    torch.cuda.empty_cache()
    return res


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes of this script (one per GPU, rendezvous on
    127.0.0.1), relay rank 0's stdout (the ONE JSON line), send the other ranks' stdout to stderr, and exit non-zero if any
    rank fails.  Called before this process has made any GPU call (a process that has initialised the GPU must never be
    replaced or forked on this pool: the children are brand-new interpreters)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), UDAPOSE_BENCH_CHILD="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=(None if r == 0 else sys.stderr)))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    for q in pending:           # one rank died: the others would wait in a collective for ever
                        q.terminate()
            time.sleep(0.2)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


class _HangGuard:
    """Data-parallel runs only.  A collective that never completes (a rank that died, a captured collective replayed out of step) would keep
    the whole job waiting until the launcher's own limit - and collectives replayed from a hipGraph are invisible to torch.distributed's
    watchdog and its timeout.  Every phase that issues collectives arms this timer with a generous bound for that phase; if it fires the rank
    says which phase hung, dumps its Python stacks and exits (code 3; 0 when the JSON line is already out), so the job ends in minutes
    with a message instead of hanging the node."""

    def __init__(self):
        self.timer, self.enabled = None, False

    def arm(self, seconds, what, code=3):
        self.cancel()
        if not self.enabled:
            return
        import faulthandler
        import threading

        def fire():
            sys.stderr.write(f"bench.py: '{what}' did not finish within {seconds:.0f} s on rank {os.environ.get('RANK', '0')} - giving up\n")
            sys.stderr.flush()
            faulthandler.dump_traceback(all_threads=True)
            os._exit(code)
        self.timer = threading.Timer(seconds, fire)
        self.timer.daemon = True
        self.timer.start()

    def cancel(self):
        if self.timer is not None:
            self.timer.cancel()
            self.timer = None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--spinup", type=float, default=8.0, help="seconds of untimed load before the warm-up steps (device clock ramp)")
    ap.add_argument("--batch", type=int, default=32, help="images per GPU per domain (BASELINE: 32)")
    ap.add_argument("--arch", default="pose_resnet101")
    ap.add_argument("--image-size", type=int, default=256, help="other configs (BASELINE.json configs[4]: 384); default = the metric's 256")
    ap.add_argument("--keypoints", type=int, default=16, help="other configs (configs[4]: 18)")
    ap.add_argument("--sigma", type=float, default=2, help="label / rectify sigma (configs[4]: 1.0)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16"], help="storage / MFMA element type: bf16 = the metric; fp16 = the "
                    "reference's autocast dtype (BASELINE.json configs[4]), with GradScaler-style loss scaling kept on the device")
    ap.add_argument("--precision", default=None, choices=["reference", "reference_fp32"], help="not the metric: 'reference' = the reference's own precision mix "
                    "(train_human.py:346-358,414): student fp16 + device-side GradScaler, teacher and style network in the fp32-grade f16x2 mode; "
                    "'reference_fp32' = the same with the exact-fp32 MFMA forms for the teacher and the style network (A/B of the f16x2 mode)")
    ap.add_argument("--strong", action="store_true", help="not the metric: strong scaling, global batch fixed at --batch (what the reference's "
                    "nn.DataParallel does): every rank takes batch / world images per domain")
    ap.add_argument("--config2", action="store_true", help="not the metric: BASELINE.json configs[2] = the same step + AdaIN s2t / t2s style "
                    "passes (both forced on, alpha 0.5, randomly initialised VGG / decoder) + adaptive occlusion; runs eagerly (the occlusion "
                    "reads confidences back to the host, as the reference does)")
    ap.add_argument("--host-inputs", action="store_true", help="not the metric: every step copies its batch from pinned host memory "
                    "(PCIe-inclusive rate, DESIGN.md section 5)")
    ap.add_argument("--igemm-tile", type=int, default=-1, help="tuning: force one igemm tile configuration id")
    ap.add_argument("--wgrad-group", type=int, default=1, help="tuning: 0 = one weight-gradient launch per layer")
    ap.add_argument("--bn-bwd-fused", type=int, default=1, help="tuning: 0 = separate BN-backward reduce launches")
    ap.add_argument("--wgrad-stages", type=int, default=0, help="tuning: 64-pixel stages per work-group of the grouped wgrad")
    ap.add_argument("--policy", action="append", default=[], metavar="FIELD=INT", help="tuning: override one field of the dispatch policy "
                    "(include/udapose.h udapose_policy), e.g. --policy igemm_h3=0; repeatable")
    ap.add_argument("--no-merge-wgrad", action="store_true", help="tuning: each pass launches its own grouped weight gradients")
    ap.add_argument("--two-graphs", action="store_true", help="tuning: the optimizer tail as its own graph on one rank too")
    ap.add_argument("--force-overlap", action="store_true", help="tuning: the data-parallel backward (two parts, gradient sums per part) without a "
                    "process group: isolates what the cut costs on one rank")
    ap.add_argument("--stream-priority", type=int, default=0, help="tuning: priority of the three branch streams (-1 = high; side streams stay 0)")
    ap.add_argument("--no-sum-in-tail", action="store_true", help="tuning: a separate launch adds the two passes' gradient buffers")
    ap.add_argument("--no-fuse-rectify", action="store_true", help="tuning: activations and rectify as two arg-max sweeps")
    ap.add_argument("--no-fuse-tail", action="store_true", help="tuning: separate Adam / EMA / weight-pack launches instead of the fused tail")
    ap.add_argument("--dp-form", default="auto", choices=["auto", "fixed"], help="data parallel (world > 1): 'auto' times 5 steps of each form - {two gradient "
                    "buckets, the first under backward part 2 | one bucket after the whole backward} x {fp32, bf16 on the wire} - during spin-up, "
                    "all ranks agree on the fastest (rank 0's choice, broadcast) and the timed region runs it; 'fixed' (or an explicit --grad-comm / "
                    "--one-bucket) runs the form the flags name")
    ap.add_argument("--grad-comm", default=None, choices=["fp32", "bf16"], help="data parallel: wire format of the gradient buckets (bf16: one "
                    "rounding per contribution, all-to-all + fp32 accumulation on the shard's owner + all-gather: half the bytes per xGMI link)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-trained-parity", action="store_true", help="skip the parity leg on a network trained in this run (~20 s)")
    ap.add_argument("--capture-comm", action="store_true", help="data parallel: capture the RCCL collectives INTO the step's one graph (round 5; +0.4 ms instead of +0.8 ms "
                    "over the plain step at one rank) instead of keeping them eager between four graphs.  Opt-in (round 6, ADVICE r5): the capture rests on a timed "
                    "wait for RCCL's watchdog and replayed collectives are invisible to torch.distributed's timeout; if it fails on any rank all ranks fall back together")
    ap.add_argument("--one-bucket", action="store_true", help="data parallel: the whole backward with one merged weight-gradient tail, then ONE "
                    "all-reduce of the flat gradient buffer (no overlap with backward part 2)")
    ap.add_argument("--split-graphs", action="store_true", help="cut the step into three graphs around the collectives even on one rank")
    ap.add_argument("--eager", action="store_true", help="launch every kernel from the host instead of replaying hipGraphs")
    ap.add_argument("--cpu-images", type=int, default=0, help="batch of the CPU baseline step (0: 8, or what fits the time budget)")
    ap.add_argument("--dp-segments", action="store_true", help="tuning: after the timed region, 20 more steps with HIP events between the parts of the "
                    "data-parallel step (graphs and collectives); device ms per part on stderr")
    ap.add_argument("--arena-teacher", action="store_true", help="tuning: no-grad forwards keep the bump-allocated 2.8 GB arena instead of the forward-only "
                    "plan's six rotating scratch buffers (A/B of round 5's default)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the untimed extra legs (fp16 / reference precision mix rates)")
    args = ap.parse_args()
    dp_auto = args.dp_form == "auto" and args.grad_comm is None and not args.one_bucket and not args.eager and not args.split_graphs
    if args.grad_comm is None:
        args.grad_comm = "fp32"

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args.gpus))            # (nothing above this line touches the GPU)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with `python bench.py --gpus {args.gpus}` (it starts its "
                         f"own ranks) or `python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 "
                         f"bench.py --gpus {args.gpus} ...`")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback); the CPU oracle lives in oracle/ and is only the baseline leg")
    # test hook (tests/test_gpu_hotpath.py): UDAPOSE_BENCH_SHARE_GPU=1 puts every rank on cuda:0 with the gloo backend, so
    # the multi-rank control flow (three graphs around the two collectives) can be exercised on a one-GPU box
    share = os.environ.get("UDAPOSE_BENCH_SHARE_GPU", "0") == "1"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    force_dist = os.environ.get("UDAPOSE_FORCE_DIST", "0") == "1"     # test hook: one-rank RCCL group, data-parallel code path
    # RCCL prints its version banner on stdout when the communicator is created (NCCL_DEBUG=VERSION in this image): stdout is
    # pointed at stderr while the process group comes up, so that rank 0's stdout carries the ONE JSON line and nothing else
    guard = _HangGuard()
    guard.enabled = world > 1 or force_dist
    guard.arm(300, "process group / communicator creation")
    t_comm0 = time.perf_counter()
    sys.stdout.flush()
    saved_out = os.dup(1)
    os.dup2(2, 1)
    try:
        if force_dist and world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29517")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if share:
                dist.init_process_group("gloo", rank=rank, world_size=world)
            else:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        if dist.is_initialized():
            t_init = torch.zeros(1, device=dev)
            dist.all_reduce(t_init)                 # first collective: communicator creation (and its banner) happens here at the latest
            torch.cuda.synchronize()
    finally:
        sys.stdout.flush()
        os.dup2(saved_out, 1)
        os.close(saved_out)
    setup_s = {"communicator_s": round(time.perf_counter() - t_comm0, 2)} if dist.is_initialized() else {}

    from uda_poseestimation_amd import _hip, synthetic
    from uda_poseestimation_amd.engine import GraphedTrainStep, MeanTeacherTrainer
    import uda_poseestimation_amd.lib.models as models
    if args.arena_teacher:
        from uda_poseestimation_amd.lib.models.pose_resnet import PoseResNet
        PoseResNet.fwd_only_plans = False
    if args.precision in ("reference", "reference_fp32"):
        args.dtype = "fp16"                # (the student's element type: the library the roofline sample's profiler hooks live in)
    lib = _hip.lib(args.dtype)
    # tuning flags -> explicit dispatch policy of both networks' executor plans (udapose_policy; empty = production policy)
    tune = {}
    if args.wgrad_group != 1:
        tune["wgrad_group"] = args.wgrad_group
    if args.wgrad_stages > 0:
        tune["wgrad_stages"] = args.wgrad_stages
    if args.bn_bwd_fused != 1:
        tune["bn_bwd_fused"] = args.bn_bwd_fused
    if args.igemm_tile >= 0:
        tune["igemm_tile"] = args.igemm_tile
    for kv in args.policy:
        k, v = kv.split("=")
        tune[k] = int(v)

    N, K = args.batch, args.keypoints
    if args.strong:
        if N % world:
            raise SystemExit(f"--strong: batch {N} is not divisible by {world} ranks")
        N //= world
    S = args.image_size
    sigma = int(args.sigma) if float(args.sigma).is_integer() and args.sigma >= 2 else float(args.sigma)
    torch.manual_seed(0)
    student = models.__dict__[args.arch](num_keypoints=K, pretrained_backbone=False).to(dev)
    teacher = models.__dict__[args.arch](num_keypoints=K, pretrained_backbone=False).to(dev)
    student.policy.update(tune)
    teacher.policy.update(tune)
    extra = {}
    if args.config2:
        extra = style_extras(dev, args.precision)
    trainer = MeanTeacherTrainer(student, teacher, lr=1e-4, teacher_alpha=0.999, lambda_c=1.0, mask_ratio=0.5, sigma=sigma, image_size=S,
                                 heatmap_size=S // 4, precision=(args.precision or args.dtype), grad_comm=args.grad_comm, **extra)
    if args.no_fuse_tail:
        trainer.fuse_tail = False
    if args.no_fuse_rectify:
        trainer.fuse_rectify = False
    trainer.stream_priority = args.stream_priority
    if args.no_merge_wgrad:
        trainer.merge_wgrad = False
    if args.two_graphs:
        trainer.single_graph = False
    if args.force_overlap:
        trainer.overlap_allreduce = True
    if args.one_bucket:
        trainer.overlap_allreduce = False
    if args.no_sum_in_tail:
        trainer.sum_grads_in_tail = False
    # configs[2] captured: the occlusion decisions are taken on the device (four uniform draws per sample, no read-back);
    # --eager keeps the reference's host draws (one read-back of the confidences per step)
    trainer.device_occlusion = bool(args.config2 and not args.eager)
    b = synthetic.mean_teacher_batch(N, num_keypoints=K, image_size=S, heatmap_size=S // 4, sigma=sigma, seed=rank)   # one shard per rank
    g = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()}

    def eager_step():
        return trainer.train_step(g["x_s"], g["label_s"], g["weight_s"], g["x_t_stu"], g["x_t_tea"], g["aug_param_stu"], g["aug_param_tea"])

    dp_choice = None
    if args.eager:
        step = eager_step
    else:
        # the whole step (~2500 kernels) is captured once into hipGraphs; every replay recomputes the re-warp matrices from
        # the batch's aug_param tuples on the host and copies them (and the batch, if it changed) into the static inputs
        def build_graphed():
            return GraphedTrainStep(trainer, g["x_s"], g["label_s"], g["weight_s"], g["x_t_stu"], g["x_t_tea"], g["aug_param_stu"],
                                    g["aug_param_tea"], split=(True if args.split_graphs else None),
                                    capture_comm=(args.capture_comm and dist.is_initialized() and dist.get_backend() == "nccl" and not args.split_graphs))

        guard.arm(420, "capture / selection of the data-parallel form")
        if dp_auto and dist.is_initialized() and (world > 1 or force_dist):
            # the ONE scaling run the driver may get should not depend on a guess about xGMI: every form is captured and timed (5 steps between
            # barrier + synchronize brackets, MAX over ranks), rank 0 picks the fastest and broadcasts the choice
            forms = [("two_buckets_fp32", True, "fp32"), ("one_bucket_fp32", False, "fp32"), ("two_buckets_bf16", True, "bf16"), ("one_bucket_bf16", False, "bf16")]
            if dist.get_backend() != "nccl":
                forms = forms[:2]                       # (bf16 on the wire needs RCCL)
            timings, built, capture_s, fallbacks = {}, {}, {}, {}
            t_sel0 = time.perf_counter()
            for name, overlap, wire in forms:
                trainer.overlap_allreduce, trainer.sync.comm_dtype = overlap, wire
                gr, err = None, None
                t_c0 = time.perf_counter()
                try:
                    gr = build_graphed()
                except Exception as e:          # (a form this stack cannot build on some rank is dropped on all of them)
                    err = e
                capture_s[name] = round(time.perf_counter() - t_c0, 2)
                if gr is not None and getattr(gr, "capture_fallback", None):
                    fallbacks[name] = gr.capture_fallback        # (the engine's collective fallback: every rank built the four-graph form instead)
                okt = torch.tensor([0.0 if gr is None else 1.0], device=dev)
                dist.all_reduce(okt, op=dist.ReduceOp.MIN)
                if float(okt.item()) < 1.0:
                    if rank == 0:
                        print(f"dp form {name}: not available ({type(err).__name__ if err else 'another rank failed'}: {err})", file=sys.stderr, flush=True)
                    timings[name] = None
                    gr = None
                    continue
                for _ in range(2):
                    gr.step(None, None, None, None, None, g["aug_param_stu"], g["aug_param_tea"])
                dist.barrier(); torch.cuda.synchronize()
                t_a = time.perf_counter()
                for _ in range(5):
                    gr.step(None, None, None, None, None, g["aug_param_stu"], g["aug_param_tea"])
                torch.cuda.synchronize()
                tt = torch.tensor([time.perf_counter() - t_a], device=dev, dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                timings[name] = round(float(tt.item()) / 5 * 1e3, 3)
                built[name] = gr
            if not built:
                raise SystemExit("bench.py: no data-parallel form could be built")
            pick = torch.tensor([min((i for i in range(len(forms)) if timings[forms[i][0]] is not None), key=lambda i: timings[forms[i][0]])],
                                device=dev, dtype=torch.int64)
            dist.broadcast(pick, 0)
            name, overlap, wire = forms[int(pick.item())]
            trainer.overlap_allreduce, trainer.sync.comm_dtype = overlap, wire
            args.grad_comm = wire
            graphed = built.pop(name)
            built.clear()
            gr = None            # (every graph that captured RCCL launches must be gone before the process group is destroyed: see the end of main)
            torch.cuda.empty_cache()
            setup_s.update({"capture_s": capture_s, "selection_s": round(time.perf_counter() - t_sel0, 2)})
            dp_choice = {"chosen": name, "ms_per_step_5_steps": timings, "collectives_captured": bool(getattr(graphed, "capture_comm", False)),
                         "capture_fallback": (fallbacks or None)}
            if rank == 0:
                print(f"dp form: {dp_choice}", file=sys.stderr, flush=True)
        else:
            t_c0 = time.perf_counter()
            graphed = build_graphed()
            if dist.is_initialized():
                setup_s["capture_s"] = {"fixed": round(time.perf_counter() - t_c0, 2)}
                if getattr(graphed, "capture_fallback", None):
                    dp_choice = {"chosen": "fixed", "ms_per_step_5_steps": None, "collectives_captured": False, "capture_fallback": {"fixed": graphed.capture_fallback}}

        host = {k: v.cpu().pin_memory() for k, v in g.items() if torch.is_tensor(v)} if args.host_inputs else None

        def step():
            if host is None:
                # inputs resident in HBM: the captured step's static input buffers ARE the resident batch (handing over other device
                # tensors would add five device-to-device copies, 84 MB, per step); the aug_param values are staged every step
                return graphed.step(None, None, None, None, None, g["aug_param_stu"], g["aug_param_tea"])
            # host batches: this step consumes the batch staged during the previous one, and the next batch's H2D copies
            # are started on the copy stream so that they run under this step's replay
            if not graphed._have_staged:
                graphed.prefetch(host["x_s"], host["label_s"], host["weight_s"], host["x_t_stu"], host["x_t_tea"])
            out = graphed.step(None, None, None, None, None, g["aug_param_stu"], g["aug_param_tea"])
            graphed.prefetch(host["x_s"], host["label_s"], host["weight_s"], host["x_t_stu"], host["x_t_tea"])
            return out

    # Device spin-up (untimed, before the W warm-up steps): an idle MI355X needs ~2-3 s of sustained load to reach its
    # steady clocks (measured: 770 img/s in a cold first run vs 970 img/s after 2.5 s of load, same binary, same box).
    # Every rank must run the SAME number of spin-up steps (each step issues collectives): the stop decision is rank 0's clock,
    # shared with a one-element all-reduce after every step (a rank-local clock test lets ranks disagree by one step at the
    # boundary, after which the collectives of the timed region no longer pair up).
    guard.arm(240 + 2 * args.spinup + 0.5 * (args.warmup + args.steps), "spin-up, warm-up and the timed steps")
    t_spin = time.perf_counter()
    spin_ms = []
    go = torch.zeros(1, device=dev) if dist.is_initialized() else None
    while True:
        more = time.perf_counter() - t_spin < args.spinup
        if go is not None:
            go.fill_(1.0 if (more and rank == 0) else 0.0)
            dist.all_reduce(go)
            more = bool(go.item() > 0)
        if not more:
            break
        t_a = time.perf_counter()
        out = step()
        torch.cuda.synchronize()
        spin_ms.append((time.perf_counter() - t_a) * 1e3)
    if spin_ms and rank == 0:
        k = max(1, len(spin_ms) // 10)
        print("spin-up ms/step by tenths: " + " ".join(f"{sum(spin_ms[i:i + k]) / len(spin_ms[i:i + k]):.1f}" for i in range(0, len(spin_ms), k)),
              file=sys.stderr)
    for _ in range(args.warmup):
        out = step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    trainer.sync.profile = dist.is_initialized()      # (HIP events around the communication calls of the compute stream: no host sync)
    t0 = time.perf_counter()
    for i in range(args.steps):         # the timed region: EXACTLY K steps (hipGraph replays unless --eager), nothing else
        out = step()
    torch.cuda.synchronize()
    t_rank = time.perf_counter() - t0                 # this rank's own time for the K steps (before the closing barrier)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        guard.arm(300, "the untimed legs after the timed region")
    else:
        guard.cancel()          # (one rank: the other configurations and the CPU baseline follow, minutes of work without collectives that could pair wrongly)
    comm_exposed_ms = trainer.sync.exposed_ms()
    if args.dp_segments and not args.eager:
        graphed.profile_segments = True
        for i in range(20):
            step()
        print("dp segments (device ms per part, mean of 20 steps):", graphed.segment_ms(), file=sys.stderr, flush=True)
        graphed.profile_segments = False
    trainer.sync.profile = False
    # The synchronous-loop rate (untimed extra, after the timed region): one torch.cuda.synchronize() per step, as a loop that reads
    # the loss / PCK of every iteration on the host forces (train_human.py:443 moves y_s to the CPU each iteration)
    sync_ms = []
    for i in range(min(args.steps, 40)):
        t_a = time.perf_counter()
        out = step()
        torch.cuda.synchronize()
        sync_ms.append((time.perf_counter() - t_a) * 1e3)
    sync_ms.sort()
    ms_synced = sync_ms[len(sync_ms) // 2] if sync_ms else None
    # The same loop with the read-back DEFERRED by one step (GraphedTrainStep.step_async: losses + device PCK of step i are read from a
    # pinned double buffer while step i+1 is already queued): what a loop that logs every iteration costs with this engine
    ms_deferred, last_metrics = None, None
    if not args.eager and host is None and getattr(graphed, "metrics", False):
        nd = min(args.steps, 40)
        torch.cuda.synchronize()
        t_a = time.perf_counter()
        for i in range(nd):
            last_metrics = graphed.step_async(None, None, None, None, None, g["aug_param_stu"], g["aug_param_tea"])
        last_metrics = graphed.flush_metrics()
        ms_deferred = (time.perf_counter() - t_a) / nd * 1e3
    # Roofline sample, UNTIMED, after the timed region: one eager step on ONE stream with HIP events recorded on the launch
    # stream around every convolution launch (events cannot be placed inside a replayed graph; single-stream so that the
    # per-launch durations are comparable with rocprofv3's, which serialises).  A dry run first creates the profiler's event
    # pool (~1100 hipEventCreate calls).
    prof = (ctypes.c_double * 9)()
    trainer.concurrent = False
    for dry in (True, False):
        lib.udapose_prof_begin()
        out_prof = eager_step()
        torch.cuda.synchronize()
        _hip.check(lib.udapose_prof_end(prof), "prof_end")
    trainer.concurrent = True
    torch.cuda.synchronize()
    rank_ms = None
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # per-rank diagnostics for a SCALE line: every rank's own ms per step and the time its compute stream waited in communication
        mine = torch.tensor([t_rank / args.steps * 1e3, comm_exposed_ms if comm_exposed_ms is not None else -1.0], device=dev, dtype=torch.float64)
        every_t = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every_t, mine)
        rank_ms = [[round(float(e[0]), 3), round(float(e[1]), 3)] for e in every_t]
    loss = float(out["loss_all"])
    assert loss == loss, "loss is NaN"
    in_sync = None
    if world > 1:
        # data-parallel invariant (outside the timed region): every rank holds the same student and teacher after the run
        chk = torch.stack([sum(p.detach().double().sum() for p in m.parameters()) for m in (student, teacher)]).to(dev)
        every = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(every, chk)
        in_sync = all(bool(torch.equal(e, every[0])) for e in every)

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = world * N * args.steps / elapsed
        fl, ms_f, fp_f = prof[0], prof[1], prof[2]
        dl, ms_d, fp_d = prof[3], prof[4], prof[5]
        wl, ms_w, fp_w = prof[6], prof[7], prof[8]
        ig_l, ig_ms, ig_fl = fl + dl, ms_f + ms_d, fp_f + fp_d
        achieved = ig_fl / (ig_ms * 1e-3) / 1e12 if ig_ms > 0 else 0.0
        layers = {"pose_resnet101": [3, 4, 23, 3], "pose_resnet50": [3, 4, 6, 3]}[args.arch]
        if args.eager:
            launch_desc = "eager"
        else:
            if world > 1 or args.split_graphs or force_dist:
                if getattr(graphed, "capture_comm", False):
                    what = ("1 hipGraph with the RCCL collectives captured inside (confidence all-gather, gradient all-reduce "
                            + ("in two buckets, the first under backward part 2)" if trainer._overlap() else "in one bucket after the whole backward)"))
                elif graphed.g_lb2 is not None:
                    what = "4 hipGraphs around the RCCL collectives (gradient all-reduce in two buckets, the first under backward part 2)"
                else:
                    what = "3 hipGraphs around the two RCCL collectives"
            else:
                what = "1 hipGraph (forwards, losses, backward, Adam + EMA + packs)" if graphed.one_graph else "2 hipGraphs"
            launch_desc = (("2 style-transfer hipGraphs (alpha on the device) + " if args.config2 else "") + what
                           + "; timed region = graph replays only (the instrumented eager roofline sample runs after it, untimed)")
        res = {
            "metric": f"images/sec (student+teacher step) {S}x{S} b={N}", "value": round(value, 2), "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "ms_per_step_synced": round(ms_synced, 3) if ms_synced is not None else None,
            "ms_per_step_deferred_readback": round(ms_deferred, 3) if ms_deferred is not None else None,
            "pck_source_batch": (round(last_metrics["acc_s"], 4) if last_metrics else None),
            "spinup_s": args.spinup, "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None,
            "dtype": ("fp16 (student) + f16x2 fp32-grade (teacher, style)" if args.precision == "reference" else
                      ("fp16 (student) + exact fp32 MFMA (teacher, style)" if args.precision == "reference_fp32" else args.dtype)), "data": "synthetic",
            "config": {"workload": f"{args.arch} K={K} mean-teacher step (student fwd+bwd on 2x{N}, teacher fwd on {N}, JointsMSE+Cons, "
                                   f"Adam, EMA), {S}x{S}, b={N}/GPU, " + ("AdaIN s2t + t2s style passes and adaptive occlusion (BASELINE.json configs[2]; NOT the metric)" if args.config2 else "no AdaIN" + (" (BASELINE.json configs[1])" if (S, K, N, args.arch, args.dtype) == (256, 16, 32, "pose_resnet101", "bf16") else
                                                    (f", K={K}, {args.dtype} (BASELINE.json configs[4] shape and dtype on one GPU; NOT the metric)"
                                                     if (S, K, args.arch, args.dtype) == (384, 18, "pose_resnet101", "fp16") else f", K={K}, {args.dtype}"))),
                       "global_batch": world * N, "parallelism": f"dp{world}",
                       **({"precision": "reference mix (train_human.py:346-358,414): student fp16 + device-side GradScaler, teacher"
                                        + (", style network" if args.config2 else "") + (" in the fp32-grade f16x2 mode" if args.precision == "reference" else
                                                                                          " in the exact-fp32 MFMA mode") + "; NOT the metric"}
                          if args.precision else {})},
            "loss": loss, "launch": launch_desc,
            "rccl_ranks": (dist.get_world_size() if (dist.is_initialized() and dist.get_backend() == "nccl") else 0),
            "grad_comm": args.grad_comm if dist.is_initialized() else None,
            "dp_form": dp_choice, "dp_setup_s": (setup_s or None),
            "comm_exposed_ms_per_step": round(comm_exposed_ms, 3) if comm_exposed_ms is not None else None,
            "rank_ms_per_step_min_max": ([min(r[0] for r in rank_ms), max(r[0] for r in rank_ms)] if rank_ms else None),
            "rank_comm_exposed_ms": ([r[1] for r in rank_ms] if rank_ms else None),
            "replicas_in_sync": in_sync, "inputs": "pinned host memory: every step's batch is copied H2D on a copy stream under the previous step" if args.host_inputs else "resident in HBM",
            "step_tflops_per_gpu": round(7 * N * FWD_GFLOP_PER_IMAGE / 1e3 / (ms * 1e-3), 2) if (args.arch, S, K) == ("pose_resnet101", 256, 16) else None,
            "policy_overrides": (tune or None), "valid": True,
            "roofline": {"bound": "mfma", "kernel": f"igemm_kernel (implicit-GEMM conv fprop+dgrad, {args.dtype} MFMA 16x16x32)",
                         "bound_note": "priced against the dense MFMA peak as SURVEY.md 8(d) prescribes for the convolutions (>99 % of the FLOPs); the MEASURED "
                                       "limiter of these launches is neither roof: per-CU L2->LDS operand fill (137-146 GB/s per CU, shared by the step's three "
                                       "streams) and launch-chain latency (MFMA pipe 4-22 % busy, 44-80 % of wave cycles in s_waitcnt / s_barrier: "
                                       "profiles/r5_pmc_sq_by_shape.txt, r4_probes.txt, r5_ab_runs.txt); hbm_frac is the same launches priced against HBM",
                         "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": measured_traffic_per_igemm_launch(),
                         "traffic_note": "HBM bytes per igemm launch from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes "
                                         f"(profiles/{os.path.basename(_pmc_traffic_file())}), not collected live",
                         "launches_per_step": int(ig_l), "avg_launch_us": round(ig_ms * 1e3 / max(ig_l, 1), 2),
                         "flops_per_launch_avg": ig_fl / max(ig_l, 1), "kernel_ms_per_step": round(ig_ms, 3),
                         "wgrad": {"launches_per_step": int(wl), "kernel_ms_per_step": round(ms_w, 3),
                                   "achieved": round(fp_w / (ms_w * 1e-3) / 1e12, 2) if ms_w > 0 else None}},
        }
        per_step = measured_traffic_per_step()
        if per_step and (args.arch, S, K, N, args.dtype, args.precision) == ("pose_resnet101", 256, 16, 32, "bf16", None) and not args.config2:      # (the PMC passes are the headline's)
            tot = sum(per_step.values())
            res["roofline"]["step_hbm"] = {"bytes_per_step": tot, "by_family": per_step, "achieved_TBps": round(tot / (ms * 1e-3) / 1e12, 3),
                                           "peak_TBps": 8.0, "note": "whole-step HBM traffic of the conv / BN / weight-gradient kernels (PMC passes under "
                                           "profiles/) over this run's step time"}
            res["roofline"]["hbm_frac"] = round(tot / (ms * 1e-3) / 8e12, 4)       # whole-step HBM bytes / step time / 8 TB/s
            # the step against the floor of its own traffic: the PMC bytes above + the optimizer sweep's 46 B / parameter + ~0.6 GB of
            # conversions / packs at the 6.0 TB/s this chip streams through the paths these kernels use, + the L2-hit part of the operand fills
            # (~66 GB of the implicit GEMMs' ~90 GB, ~0.5 ms for the weight gradients) at the 35 TB/s that path gives - all constants cited from profiles/
            floor_ms = (tot + 2.5e9 + 0.6e9) / 6.0e12 * 1e3 + 66e9 / 35e12 * 1e3 + 0.5
            res["roofline"]["step_hbm"]["traffic_floor"] = {
                "floor_ms": round(floor_ms, 2), "frac_of_floor": round(floor_ms / ms, 4), "cited": True,
                "note": "what this step's own memory traffic costs with every latency hidden (57 GB at 6.0 TB/s + L2-hit operand fills at 35 TB/s; "
                        "profiles/r4_probes.txt, r5_probe_lds_fill_tiles.txt, r5_pmc_hbm_traffic.txt) over this run's step time: the statement that goes "
                        "with 'frac' (the convolutions priced against the MFMA peak)"}
        # the CPU oracle leg (bounded sample) and, from its first step, the MEASURED parity of this run's precision(s)
        par_ok = (S, K) == (256, 16) and not args.config2
        if not args.no_cpu_baseline and world == 1:     # (the CPU leg runs on rank 0 of the ONE-rank run only; N > 1 lines carry null)
            try:
                res["cpu_baseline"] = cpu_baseline(args.cpu_images, layers)
            except Exception as e:
                res["cpu_baseline"] = {"value": None, "unit": "images/sec", "cores": 0, "kind": "port", "sample": f"failed: {type(e).__name__}: {e}"}
            if par_ok and res["cpu_baseline"]["value"] is not None:
                try:
                    del graphed
                except NameError:
                    pass
                torch.cuda.empty_cache()
                # (the parity legs never cost the line its headline: a failure is reported in their place)
                print("parity: device step from the oracle's start state ...", file=sys.stderr, flush=True)
                try:
                    res["parity"] = measured_parity(args.arch, dev, args.precision or args.dtype)
                except Exception as e:
                    res["parity"] = {"measured": False, "error": f"{type(e).__name__}: {e}"}
                if not args.no_trained_parity:
                    print("parity: training the trained-like network, one oracle step from it ...", file=sys.stderr, flush=True)
                    try:
                        ctx_tr = trained_parity_context(args.arch, res["cpu_baseline"]["cores"])
                        res["parity_trained"] = measured_parity(args.arch, dev, args.precision or args.dtype, ctx_tr)
                    except Exception as e:
                        ctx_tr = None
                        res["parity_trained"] = {"measured": False, "error": f"{type(e).__name__}: {e}"}
        elif world > 1:
            res["cpu_baseline"] = None
        # the parity-compliant configurations of the same step, driver-visible (untimed extras after the headline's timed region)
        headline_cfg = (args.arch, S, K, N, args.dtype, args.precision) == ("pose_resnet101", 256, 16, 32, "bf16", None)
        if world == 1 and headline_cfg and not (args.config2 or args.eager or args.no_other_configs or args.host_inputs or tune):
            try:
                del graphed
            except NameError:
                pass
            torch.cuda.empty_cache()
            oc = {}
            for tag, dt_, pr_ in (("fp16", "fp16", None), ("reference_mix", "fp16", "reference")):
                print(f"other_configs: {tag} ...", file=sys.stderr, flush=True)
                try:
                    oc[tag] = other_config_rate(args.arch, dev, N, K, S, sigma, dt_, pr_)
                except Exception as e:
                    oc[tag] = {"error": f"{type(e).__name__}: {e}"}
                    continue
                # (measured by THIS run, like res["parity"]: the device step in that precision against the cpu_baseline leg's first oracle step)
                try:
                    oc[tag]["parity"] = measured_parity(args.arch, dev, pr_ or dt_) if (res.get("parity") or {}).get("measured") else None
                    oc[tag]["parity_trained"] = measured_parity(args.arch, dev, pr_ or dt_, ctx_tr) if (res.get("parity_trained") or {}).get("measured") else None
                except Exception as e:
                    oc[tag]["parity_error"] = f"{type(e).__name__}: {e}"
            oc["note"] = ("`parity` = this run's measurement on the bench's own randomly initialised network (train-mode BN: every bottleneck amplifies storage "
                          "rounding; no 16-bit format is close to the fp32 oracle there), `parity_trained` = the same on a network trained in this run; "
                          "the benchmarked size (N = 32, captured) against the oracle is tests/test_gpu_fullsize.py")
            # BASELINE.json's other single-GPU configurations (the ones the reference actually trains: train_human.py:345-358,
            # train_animal.py:330-483), same harness: 20 graph replays each
            for tag, kw in (("configs[2]_bf16_style", dict(dtype="bf16", precision=None, config2=True)),
                            ("configs[2]_reference_mix", dict(dtype="fp16", precision="reference", config2=True)),
                            ("configs[4]_workload_1gpu_fp16", dict(dtype="fp16", precision=None, K=18, S=384, sigma=1.0))):
                print(f"other_configs: {tag} ...", file=sys.stderr, flush=True)
                kw = dict(kw)
                try:
                    oc[tag] = other_config_rate(args.arch, dev, N, kw.pop("K", K), kw.pop("S", S), kw.pop("sigma", sigma), kw.pop("dtype"), kw.pop("precision"), **kw)
                except Exception as e:
                    oc[tag] = {"error": f"{type(e).__name__}: {e}"}
            oc["configs[2]_bf16_style"]["workload"] = "configs[1] + AdaIN s2t and t2s style passes (both forced on, alpha 0.5, seeded random VGG / decoder) + adaptive occlusion; 2 style hipGraphs + 1 step hipGraph"
            oc["configs[2]_reference_mix"]["workload"] = "the same in the reference's precision mix (fp16 student, f16x2 teacher and style network)"
            oc["configs[4]_workload_1gpu_fp16"]["workload"] = f"{args.arch} K=18, 384x384 (heat-maps 96x96), sigma 1.0, fp16, b={N} on ONE GPU (configs[4] is this workload on 8)"
            res["other_configs"] = oc
        print(json.dumps(res), flush=True)
    if os.environ.get("UDAPOSE_BENCH_DEBUG_EXIT"):
        import faulthandler
        faulthandler.dump_traceback_later(25, exit=True)
    guard.arm(90, "process group teardown (the result line is already printed)", code=0)
    if world > 1:
        dist.barrier()
    if dist.is_initialized():
        # graphs that hold captured RCCL launches go first: RCCL's communicator teardown waits for every graph that captured its kernels to be
        # destroyed (measured: destroy_process_group() hangs for ever with one such graph still referenced)
        graphed = built = None
        import gc
        gc.collect()
        torch.cuda.synchronize()
        dist.destroy_process_group()
    guard.cancel()


if __name__ == "__main__":
    main()
