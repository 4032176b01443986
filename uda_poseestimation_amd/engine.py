"""Mean-teacher training step of the reference (train_human.py:244-302 `pretrain`, :305-458 `train`) on the MI355X path.

The reference keeps this logic inside its train scripts; this harness reproduces the order of operations so that
bench.py, smoke() and the tests can drive the hot path the way `train_human.py` does:

    zero_grad -> [style s2t / t2s] -> teacher forward (no grad, train-mode BN) -> re-warp teacher heat-maps
    -> [occlusion] -> student(x_s), student(x_t_stu) (two separate forwards) -> re-warp student heat-maps
    -> JointsMSE(source) -> activations / rectify / k-th value mask -> ConsLoss -> backward
    -> (data parallel: one all-reduce of the flat gradient buffer over RCCL) -> Adam -> EMA -> PCK

Data parallelism: one process per GPU, the batch is sharded per image, every rank holds full
student / teacher replicas; the only collectives are the gradient all-reduce (student only) and a tiny all-gather of
the per-key-point confidences so that the mask threshold stays the GLOBAL-batch k-th value.  BN statistics stay per
rank, as with the reference's nn.DataParallel.
"""
import numpy as np
import os

import torch
import torch.distributed as dist

from . import optim as fused_optim
from . import utils as mt
from . import warp
from .lib import keypoint_detection as kd
from .lib.models.loss import ConsLoss, JointsMSELoss


def _dist_on():
    # UDAPOSE_FORCE_DIST=1 (test hook): a one-rank process group still takes the data-parallel path, so the real RCCL
    # collectives and the three-graph step can be exercised on a one-GPU box
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("UDAPOSE_FORCE_DIST", "0") == "1"


class GradSync:
    """All-reduce (mean) of the student's flat fp32 gradient buffer over RCCL/xGMI.  (The PoseResNet executor writes every
    parameter gradient into views of a single buffer.)  Either ONE collective per step (`sync()`), or two buckets so that
    the first - the suffix holding layer3 / layer4 / upsampling / head, 94 % of the bytes, final when backward part 1 has
    run - travels under backward part 2: `start_upper()` right after part 1, `finish()` after part 2."""

    def __init__(self, model, comm_dtype="fp32"):
        self.model = model
        self._work, self._upper, self._cstream = None, None, None
        # 'bf16': the buckets travel in bf16 (udapose_comm_*: one rounding per contribution, all-to-all of shards, fp32 accumulation on
        # the shard's owner, the MEAN rounded to bf16 once more for the all-gather of the averaged shards): half the bytes per xGMI
        # link, two bf16 roundings per averaged gradient element.  RCCL backend only.
        self.comm_dtype = comm_dtype
        self._cbuf = {}
        # timing (bench.py): HIP events around finish() on the current stream = what the step WAITS for communication
        self.profile = False
        self._events = []

    def _flat(self):
        flat = getattr(self.model, "_flat_grad", None)
        if flat is None:
            raise RuntimeError("GradSync: model has no flat gradient buffer yet (run backward first)")
        return flat

    def _reduce_bf16(self, t, tag):
        """Average the fp32 bucket `t` over the ranks with bf16 on the wire (synchronous with respect to the current stream)."""
        from ._hip import check, lib, ptr, stream
        if dist.get_backend() != "nccl":
            raise RuntimeError("GradSync(comm_dtype='bf16') needs the RCCL backend")
        w, n = dist.get_world_size(), t.numel()
        m = (n + w - 1) // w
        m = (m + 7) // 8 * 8
        buf = self._cbuf.get(tag)
        if buf is None or buf[0].numel() != w * m or buf[0].device != t.device:
            buf = (torch.empty(w * m, dtype=torch.bfloat16, device=t.device), torch.empty(w * m, dtype=torch.bfloat16, device=t.device),
                   torch.empty(m, dtype=torch.bfloat16, device=t.device))
            self._cbuf[tag] = buf
        send, recv, mine = buf
        check(lib().udapose_comm_pack_bf16(stream(), ptr(t), n, ptr(send), w * m), "comm_pack_bf16")
        dist.all_to_all_single(recv, send)
        check(lib().udapose_comm_shard_mean(stream(), ptr(recv), w, m, ptr(mine)), "comm_shard_mean")
        dist.all_gather_into_tensor(send, mine)
        check(lib().udapose_comm_unpack_bf16(stream(), ptr(send), ptr(t), n), "comm_unpack_bf16")

    def _reduce(self, t, async_op=False, tag="all"):
        if self.comm_dtype == "bf16":
            self._reduce_bf16(t, tag)
            return None
        if dist.get_backend() == "nccl":
            return dist.all_reduce(t, op=dist.ReduceOp.AVG, async_op=async_op)   # RCCL averages in the collective: no second sweep
        w = dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=async_op)          # gloo (CPU tests, shared-GPU test) has no AVG
        if not async_op:
            t.mul_(1.0 / dist.get_world_size())
        return w

    def __call__(self):
        if not _dist_on():
            return
        ev = self._tick()
        self._reduce(self._flat())
        self._tock(ev)

    def _tick(self):
        if not self.profile:
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def _tock(self, e0):
        if e0 is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self._events.append((e0, e1))

    def exposed_ms(self):
        """Mean time per step the compute stream spent inside the communication calls (after a synchronize)."""
        if not self._events:
            return None
        v = [a.elapsed_time(b) for a, b in self._events]
        self._events = []
        return sum(v) / len(v)

    def start_upper(self):
        """Launch the all-reduce of the gradient suffix on the communicator's own stream (it waits for what is enqueued on the
        current stream - backward part 1 and the sum of the two passes - and then runs beside whatever comes next)."""
        if not _dist_on():
            return
        flat = self._flat()
        self._upper = flat[self.model.grad_split_offset():]
        if self.comm_dtype == "bf16":
            # a side stream runs the bf16 exchange of the suffix beside backward part 2 (the collectives' own streams wait for it)
            if getattr(self, "_cstream", None) is None:
                self._cstream = torch.cuda.Stream(device=flat.device)
            cur = torch.cuda.current_stream()
            self._cstream.wait_stream(cur)
            with torch.cuda.stream(self._cstream):
                self._reduce_bf16(self._upper, "upper")
            self._work = "stream"
            return
        self._work = self._reduce(self._upper, async_op=True, tag="upper")

    def finish(self):
        """All-reduce the prefix (layer2 / layer1 / stem: 6 % of the bytes), then join the suffix's collective."""
        if not _dist_on():
            return
        flat = self._flat()
        ev = self._tick()
        self._reduce(flat[:self.model.grad_split_offset()], tag="lower")
        if self._work == "stream":
            torch.cuda.current_stream().wait_stream(self._cstream)
            self._work = None
        elif self._work is not None:
            self._work.wait()                        # the current stream waits for the suffix's collective
            if dist.get_backend() != "nccl":
                self._upper.mul_(1.0 / dist.get_world_size())
            self._work = None
        self._tock(ev)


def pick_concurrent_streams(device, n=3, candidates=10):
    """`n` torch streams whose kernels demonstrably run side by side.  HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4)
    by a least-users rule that depends on every stream the process has created so far, and two streams on one queue
    serialise (tools/probe_stream_pairs.py: with 4 queues a third of all pairs do).  So the choice is MEASURED: a short spin kernel on each
    stream of a pair, started together - wall time ~T: concurrent, ~2T: same queue - over `candidates` fresh streams until `n` mutually
    concurrent ones are found (falls back to the first `n` if there is no such set).  ~50 ms, once, at capture time."""
    import itertools
    import time
    pool = [torch.cuda.Stream(device=device) for _ in range(candidates)]
    cycles = 400_000

    def wall(ss):
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for s_ in ss:
            with torch.cuda.stream(s_):
                torch.cuda._sleep(cycles)
        torch.cuda.synchronize(device)
        return time.perf_counter() - t0

    wall(pool[:1])
    t1 = min(wall(pool[:1]) for _ in range(3))
    ok = {}
    for i, j in itertools.combinations(range(candidates), 2):
        ok[(i, j)] = min(wall([pool[i], pool[j]]) for _ in range(2)) < 1.5 * t1
    for comb in itertools.combinations(range(candidates), n):
        if all(ok[p_] for p_ in itertools.combinations(comb, 2)):
            return [pool[i] for i in comb], True
    return pool[:n], False


def gather_activates(act):
    """All ranks' [N_local, K] confidences -> flat global vector (for the global-batch k-th value, train_human.py:429)."""
    if not _dist_on():
        return None
    bufs = [torch.empty_like(act) for _ in range(dist.get_world_size())]
    dist.all_gather(bufs, act.contiguous())
    return torch.cat([b.reshape(-1) for b in bufs])


class MeanTeacherTrainer:
    def __init__(self, student, teacher, lr=1e-4, teacher_alpha=0.999, lambda_c=1.0, mask_ratio=0.5, sigma=2, image_size=256,
                 heatmap_size=64, use_sgd=False, style_net=None, recover=None, s2t_freq=0.5, t2s_freq=0.5, s2t_alpha=(0.0, 1.0),
                 t2s_alpha=(0.0, 1.0), rng=None, occlude_rate=-1.0, occlude_thresh=0.9, occlude_size=10, image_px=None, precision=None,
                 loss_scale_init=65536.0, loss_scale_interval=2000, grad_comm="fp32"):
        # a single-device nn.DataParallel wrap (the reference's call form) is unwrapped: the engine drives the executor's own entry points
        # (prepare / forward_deferred_bn / finish_wgrad ...), which live on the module
        student, teacher = getattr(student, "module", student), getattr(teacher, "module", teacher)
        self.student, self.teacher = student, teacher
        self.criterion, self.con_criterion = JointsMSELoss(), ConsLoss()
        # precision: None keeps what the networks are set to (a new PoseResNet is 'auto': a differentiable forward outside
        # autocast runs bf16, the teacher's no-grad forward the fp32-grade 'f16x2' mode).
        # 'reference' = the reference's own precision mix (train_human.py:346-358,414): the student in fp16 (its autocast dtype)
        #   with GradScaler's dynamic loss scaling kept on the device (optim.py), the teacher and the style network OUTSIDE autocast
        #   in fp32 - here the fp32-grade 'f16x2' mode (three fp16 MFMAs per K step; heat-maps within ~4e-5 of the fp32 oracle).
        # 'bf16' / 'fp16': student AND teacher in that 16-bit type (BASELINE.json's benched configuration is 'bf16'; the style
        #   network keeps its own setting).
        # (the networks may arrive wrapped in a single-device nn.DataParallel, as in the reference: attributes go to the modules)
        stu_m, tea_m = getattr(student, "module", student), getattr(teacher, "module", teacher)
        if precision in ("reference", "reference_fp32"):
            # ('reference_fp32': the same mix with the EXACT fp32 MFMA forms for the teacher and the style network - the slow way to the
            # same numbers, kept for A/B timing of the f16x2 mode)
            hi = "f16x2" if precision == "reference" else "fp32"
            stu_m.precision, tea_m.precision = "fp16", hi
            sn = getattr(style_net, "module", style_net)
            if sn is not None and hasattr(sn, "precision"):
                sn.precision = hi
        elif precision is not None:
            stu_m.precision = precision
            tea_m.precision = precision
        # Loss scaling is decided ONCE, here, from what the student can run: 'fp16', or 'auto' (which resolves to fp16 whenever the step
        # runs under torch.autocast, the reference's call form) get the device-side GradScaler; a scaled bf16 step is exact (powers of
        # two), an unscaled fp16 step underflows silently - _check_scaler() refuses that combination at step time
        sp = getattr(stu_m, "precision", "bf16")
        scaled = sp in ("fp16", "auto")
        # the teacher's fp32-grade plans live in the student's library build, so the fused optimizer tail hands both plans to one .so
        if hasattr(tea_m, "aux_lib_kind"):
            tea_m.aux_lib_kind = "fp16" if sp == "fp16" else "bf16"
        sc = dict(dynamic_loss_scale=True, init_scale=loss_scale_init, growth_interval=loss_scale_interval) if scaled else {}
        if use_sgd:
            self.stu_optimizer = fused_optim.FusedSGD(student.parameters(), lr=lr, momentum=0.9, weight_decay=1e-4, nesterov=True, **sc)
        else:
            self.stu_optimizer = fused_optim.FusedAdam(student.parameters(), lr=lr, **sc)
        self.tea_optimizer = mt.OldWeightEMA(teacher, student, alpha=teacher_alpha)   # ctor copies student -> teacher
        self.sync = GradSync(student, comm_dtype=grad_comm)
        self.lambda_c, self.mask_ratio, self.sigma = lambda_c, mask_ratio, sigma
        self.ratio = image_size / heatmap_size
        self.style_net, self.recover = style_net, recover
        self.s2t_freq, self.t2s_freq, self.s2t_alpha, self.t2s_alpha = s2t_freq, t2s_freq, s2t_alpha, t2s_alpha
        self.rng = rng if rng is not None else np.random   # the reference draws from the global np.random
        self._side = None
        self.fuse_rectify = True    # activates and the rectified teacher heat-maps from one arg-max sweep (False: two launches, the reference's two calls)
        self.concurrent = True      # False: run the three branches back to back on the current stream (profiling)
        # adaptive key-point occlusion (train_human.py:374-412); rate <= -1 disables it like `--occlude-rate -1`
        self.occlude_rate, self.occlude_thresh, self.occlude_size = occlude_rate, occlude_thresh, occlude_size
        self.image_px = image_px if image_px is not None else image_size
        # False: the reference's host draws in its order (rand / choice / randint x 2 per qualifying sample, after one read-back
        # of the confidences).  True: four uniform draws per sample go to the device and udapose_occlusion_pick takes the
        # decisions there - same distribution, no read-back, capturable (GraphedTrainStep needs it)
        self.device_occlusion = False
        self.occl_rng = self.rng            # generator of the host-side occlusion draws (the reference: the global np.random)
        self._occl = None                   # ("host", aug_param_stu) | ("device", theta_back [N,1,6], u [N,4])
        # data parallel: cut the backward after layer3 and all-reduce the finished 94 % of the gradient under the rest of it
        # (None: whenever a process group is active and the network has the layer3 boundary)
        self.overlap_allreduce = None
        self.fuse_tail = True               # Adam + EMA + weight packs in one sweep (optim.FusedAdam.fused_tail_step)
        self.stream_priority = 0            # priority of the branch streams (and of a captured step's origin stream): -1 = high
        self.merge_wgrad = True             # one rank: both passes' grouped weight gradients in one launch (pose_resnet.finish_wgrad)
        self.single_graph = True            # one rank: the optimizer tail is captured into the step's graph (one launch per step)
        self.sum_grads_in_tail = True       # ... which also adds the two passes' gradient buffers (no separate axpy; one rank only)
        self.fused_last = False

    def _check_scaler(self):
        """An fp16 student forward needs the loss scaler the optimizer was built with (or the caller's own GradScaler: then build the
        trainer's optimizer yourself)."""
        hd = getattr(getattr(self.student, "module", self.student), "_last_hd", None)
        if hd is not None and hd.precision == "fp16" and getattr(self.stu_optimizer, "_scaler", None) is None:
            raise RuntimeError("the student ran in fp16 but the trainer's optimizer has no dynamic loss scaling: fp16 heat-map gradients "
                               "underflow without it - construct MeanTeacherTrainer with precision='fp16' / 'reference' / 'auto'")

    # ------------------------------------------------------------------ train_human.py:262-302
    def pretrain_step(self, x_s, label_s, weight_s, x_t=None):
        self.student.train()
        self.stu_optimizer.zero_grad()
        if self.style_net is not None and self.s2t_freq > self.rng.rand():
            a = self.rng.uniform(*self.s2t_alpha)
            x_s = self.style_net(x_s, x_t, a, clamp=self.recover)[2]
        y_s = self.student(x_s)
        loss = self.criterion(y_s, label_s, weight_s)
        self._check_scaler()
        overlap = self._overlap()
        self.student.split_backward = overlap
        self.stu_optimizer.scale_loss(loss).backward()          # (scaler.scale(loss).backward(), train_human.py:285; identity in bf16)
        self.student.split_backward = False
        if overlap:
            self.sync.start_upper()
        self._sync_grads()
        self.stu_optimizer.step()
        return {"loss_all": loss.detach(), "loss_s": loss.detach(), "y_s": y_s.detach()}

    # ------------------------------------------------------------------ train_human.py:326-444
    def train_step(self, x_s, label_s, weight_s, x_t_stu, x_t_teas, aug_param_stu, aug_params_tea, with_accuracy=False):
        """Eager step from the loader's collated batch (aug_param tuples as produced by the reference's transforms)."""
        if not isinstance(x_t_teas, (list, tuple)):
            x_t_teas, aug_params_tea = [x_t_teas], [aug_params_tea]
        n, dev = x_t_stu.shape[0], x_t_stu.device
        theta_stu = warp.recon_thetas(aug_param_stu, n, self.ratio, dev)
        thetas_tea = [warp.recon_thetas(ap, n, self.ratio, dev) for ap in aug_params_tea]
        x_s_in, x_t_teas_in = self._style_part(x_s, list(x_t_teas))
        self._occl = None
        if self.occlude_rate > -1:
            if self.device_occlusion:
                self._occl = ("device", warp.occlusion_back_thetas(aug_param_stu, n, self.ratio, dev), self.draw_occlusion_uniforms(n, dev))
            else:
                if _dist_on() and self.style_net is not None and self.occl_rng is self.rng:
                    raise RuntimeError("data parallel with style transfer: the host occlusion draws are data dependent in number and "
                                       "would take the ranks' common style decisions out of step - set trainer.device_occlusion = True "
                                       "or give the occlusion its own generator (trainer.occl_rng)")
                self._occl = ("host", aug_param_stu)
        out = self._forward_backward(x_s_in, label_s, weight_s, x_t_stu, x_t_teas_in, theta_stu, thetas_tea)
        self._sync_grads()
        self._update()
        if with_accuracy:
            _, avg_acc, cnt, _ = kd.accuracy(out["y_s"], label_s)
            out["acc_s"], out["cnt_s"] = avg_acc, cnt
        return out

    def draw_style_decisions(self):
        """The step's host draws for the style pass, in the reference's order (train_human.py:345-358): rand, [uniform], rand,
        [uniform] -> (alpha_s2t | None, alpha_t2s | None)."""
        if self.style_net is None:
            return None, None
        a_s2t = self.rng.uniform(*self.s2t_alpha) if self.s2t_freq > self.rng.rand() else None
        a_t2s = self.rng.uniform(*self.t2s_alpha) if self.t2s_freq > self.rng.rand() else None
        return a_s2t, a_t2s

    def draw_occlusion_uniforms(self, n, device=None):
        """[n,4] uniform numbers for this rank's samples.  Data parallel: every rank draws the GLOBAL batch's numbers and keeps its
        slice - with equally seeded generators the ranks' host streams stay in step (the style decisions that follow are per
        step for the whole global batch, as in the reference's single process) and the samples' draws stay independent."""
        if _dist_on():
            w, r = dist.get_world_size(), dist.get_rank()
            u = np.asarray(self.rng.rand(w * n, 4), dtype=np.float32)[r * n:(r + 1) * n]
        else:
            u = np.asarray(self.rng.rand(n, 4), dtype=np.float32)
        u = torch.from_numpy(np.ascontiguousarray(u))
        return u.to(device, non_blocking=True) if device is not None else u

    def _style_part(self, x_s, x_t_teas):
        """Bidirectional style transfer of the step's inputs (train_human.py:345-358), both from the ORIGINAL images."""
        a_s2t, a_t2s = self.draw_style_decisions()
        x_s_ori, x_t_teas_ori = x_s, list(x_t_teas)
        net = getattr(self.style_net, "module", self.style_net)          # (a single-device DataParallel wrap, as in the reference)
        with torch.no_grad():
            if hasattr(net, "encode_features") and not getattr(net, "compute_losses", False):
                # both directions transfer between the same batches: every image is encoded ONCE (the reference's two
                # style_net(...) calls encode x_s and x_t twice each; same kernels, same inputs - bit-identical results)
                if a_s2t is not None or a_t2s is not None:
                    f_s = net.encode_features(x_s_ori)
                    f_ts = [net.encode_features(x_t) for x_t in (x_t_teas_ori if a_t2s is not None else x_t_teas_ori[:1])]
                if a_s2t is not None:
                    x_s = net.transfer_from_features(f_s, f_ts[0], a_s2t, clamp=self.recover)
                if a_t2s is not None:
                    x_t_teas = [net.transfer_from_features(f_t, f_s, a_t2s, clamp=self.recover) for f_t in f_ts]
                return x_s, x_t_teas
            if a_s2t is not None:
                x_s = self.style_net(x_s_ori, x_t_teas_ori[0], a_s2t, clamp=self.recover)[2]
            if a_t2s is not None:
                x_t_teas = [self.style_net(x_t, x_s_ori, a_t2s, clamp=self.recover)[2] for x_t in x_t_teas_ori]
        return x_s, x_t_teas

    def _forward_part(self, x_s, label_s, weight_s, x_t_stu, x_t_teas, theta_stu, thetas_tea):
        """All forwards of the step after the style pass (no host reads unless the occlusion draws on the host: capturable
        in a hipGraph)."""
        student, teacher = self.student, self.teacher
        student.train()
        teacher.train()                     # the teacher's BN uses batch statistics too (train_human.py:321)
        self.stu_optimizer.zero_grad()
        # Three independent branches run on three HIP streams and meet at the consistency loss: the teacher branch (forward
        # + re-warp), the student's target-domain forward (+ re-warp) and the student's source-domain forward.  Their kernels
        # interleave on the device (measured: two concurrent fwd+bwd passes take 22.3 ms instead of 28.6 ms back to back).
        # Autograd runs each backward on the stream of its forward, so the two backward passes overlap as well; each pass
        # owns its activation arena, scratch workspace and (second pass) gradient buffer.  BN running statistics of the
        # target-domain forward are applied after the join, in the reference's call order (x_s first, then x_t_stu).
        main = torch.cuda.current_stream()
        if self._side is None or self._side[0].device != x_s.device:
            pr = self.stream_priority
            self._side = (torch.cuda.Stream(device=x_s.device, priority=pr), torch.cuda.Stream(device=x_s.device, priority=pr))
        s_tea, s_stu = self._side if self.concurrent else (main, main)
        occl = self._occl if self.occlude_rate > -1 else None
        student.prepare(x_s)                # bf16 weight packs refreshed on `main` before the branches fork
        with torch.no_grad():
            teacher.prepare(x_t_teas[0])
        s_tea.wait_stream(main)
        s_stu.wait_stream(main)
        with torch.cuda.stream(s_tea), torch.no_grad():
            y_t_teas = [teacher(x_t) for x_t in x_t_teas]
            recons = [warp.warp_chain(y, th) for y, th in zip(y_t_teas, thetas_tea)]
            y_t_tea_recon = warp.mean_views(recons)          # (k teacher views, train_human.py:361-372: one launch; k = 1: the view itself)
        if occl is not None:
            # the occlusion needs the teacher's re-warped heat-maps: the source-domain forward is issued first (it runs under
            # the teacher's), the target-domain branch waits for the teacher
            y_s = student(x_s)
            s_stu.wait_stream(s_tea)
            for t in y_t_teas + recons + [y_t_tea_recon]:
                t.record_stream(s_stu)
        with torch.cuda.stream(s_stu):
            if occl is not None:
                with torch.no_grad():
                    if occl[0] == "device":
                        x_t_stu, self.occluded = warp.occlude_keypoints_device(x_t_stu, y_t_tea_recon, theta_stu, occl[1], occl[2], self.ratio,
                                                                               self.image_px, self.occlude_rate, self.occlude_thresh,
                                                                               self.occlude_size)
                    else:   # (one small D2H of confidences, as in the reference)
                        x_t_stu, self.occluded = warp.occlude_keypoints(x_t_stu, y_t_tea_recon, occl[1], self.ratio, self.image_px,
                                                                        self.occlude_rate, self.occlude_thresh, self.occlude_size, self.occl_rng)
            y_t_stu = student.forward_deferred_bn(x_t_stu)     # separate forwards: separate BN statistics per domain
            y_t_stu_recon = warp.warp_chain(y_t_stu, theta_stu)
        if occl is None:
            y_s = student(x_s)
        main.wait_stream(s_stu)
        student.apply_deferred_bn()         # (x_s first, then x_t_stu: the reference's call order, train_human.py:414-417)
        for t in (y_t_stu, y_t_stu_recon, x_t_stu):
            t.record_stream(main)
        main.wait_stream(s_tea)
        for t in y_t_teas + recons + [y_t_tea_recon]:
            t.record_stream(main)
        with torch.no_grad():
            # activates from the heat-maps BEFORE rectify (train_human.py:427); the rectified maps (:431) come out of the same arg-max sweep
            if self.fuse_rectify:
                activates, y_t_tea_rect = mt.activations_and_rectify(y_t_tea_recon, self.sigma)
            else:
                activates, y_t_tea_rect = mt.heatmap_activations(y_t_tea_recon), None
        return {"y_s": y_s, "y_t_stu_recon": y_t_stu_recon, "y_t_tea_recon": y_t_tea_recon, "activates": activates, "y_t_tea_rect": y_t_tea_rect,
                "label_s": label_s, "weight_s": weight_s, "main": main, "s_stu": s_stu}

    def _overlap(self):
        on = self.overlap_allreduce
        if on is None:
            on = _dist_on()
        return bool(on) and hasattr(self.student, "finish_backward")

    def _sync_grads(self):
        """What sits between backward and the optimizer: nothing on one rank; with overlap, backward part 2 under the first
        bucket's all-reduce, then the second bucket; otherwise one all-reduce of the whole buffer."""
        if self._overlap() and self.student._pending_lower:
            self._backward_lower()
            self.sync.finish()
        else:
            self.sync()

    def _backward_lower(self, grads_part=2):
        student = self.student
        main = torch.cuda.current_stream()
        streams = {id(p[5]): p[5] for p in student._pending_lower}
        for stv in streams.values():
            if stv is not main:
                stv.wait_stream(main)              # (fork: inside a capture the side stream joins the graph here)
        student.finish_backward()
        for stv in streams.values():
            if stv is not main:
                main.wait_stream(stv)
        student.finish_grads(part=grads_part)

    def _loss_backward_part(self, st, gathered_activates):
        """Losses and backward from the forward state; `gathered_activates` = all ranks' confidences (None on one rank).
        With overlap (data parallel) this is backward PART 1 of both passes, the sum of their gradient suffixes and the launch
        of that suffix's all-reduce; _sync_grads() runs part 2 and the rest."""
        student = self.student
        main, s_stu = st["main"], st["s_stu"]
        overlap = self._overlap()
        student.split_backward = bool(overlap)
        # one rank: the two passes' grouped weight gradients go out as ONE launch after both gradient chains (finish_wgrad)
        merge = (not overlap) and self.merge_wgrad and hasattr(student, "finish_wgrad")
        student.merge_wgrad = bool(merge)
        if getattr(self, "_metrics_cb", None) is not None:
            self._metrics_cb(st)                # (GraphedTrainStep: decode + PCK of the source batch on an idle side stream, under the backward)
        loss_s = self.criterion(st["y_s"], st["label_s"], st["weight_s"])
        self._check_scaler()
        with torch.no_grad():
            # threshold = k-th value over the GLOBAL batch (all-gather of [N,K] floats when data parallel)
            tea_mask, _, _ = mt.confidence_mask(st["y_t_tea_recon"], self.mask_ratio, None, gathered_activates, st["activates"])
            y_t_tea_rect = st["y_t_tea_rect"] if st.get("y_t_tea_rect") is not None else mt.rectify(st["y_t_tea_recon"], sigma=self.sigma)
        loss_c = self.con_criterion(st["y_t_stu_recon"], y_t_tea_rect, tea_mask=tea_mask)
        loss_all = loss_s + self.lambda_c * loss_c
        self.stu_optimizer.scale_loss(loss_all).backward()      # (scaler.scale(loss_all).backward(), train_human.py:436; identity in bf16)
        if s_stu is not main:
            main.wait_stream(s_stu)             # the target-domain backward ran on its own stream
        student.split_backward = False
        if merge:
            student.merge_wgrad = False
            student.finish_wgrad()
        if overlap:
            student.finish_grads(part=1)    # the suffix of both passes is final: sum it ...
            if not torch.cuda.is_current_stream_capturing() or getattr(self, "capture_comm", False):
                self.sync.start_upper()     # ... and send it off (a captured step issues the collective between its graphs, or captures it: capture_comm)
        elif student._pending_lower:
            self._backward_lower(grads_part=0)      # (overlap forced off after part 1 ran: finish the backward here)
        else:
            # adds the second pass's gradient buffer (no-op when both ran on one stream) - unless the fused optimizer tail will
            # read both buffers itself (one rank: nothing else looks at the gradients in between)
            student.finish_grads(defer=self._tail_sums_grads())
        # (the re-warped heat-maps of both networks are handed out for the parity tests and bench.py's measured parity: references only)
        return {"loss_all": loss_all.detach(), "loss_s": loss_s.detach(), "loss_c": loss_c.detach(), "y_s": st["y_s"].detach(),
                "tea_mask": tea_mask, "y_t_tea_recon": st["y_t_tea_recon"], "y_t_stu_recon": st["y_t_stu_recon"].detach()}

    def _forward_backward(self, x_s, label_s, weight_s, x_t_stu, x_t_teas, theta_stu, thetas_tea):
        st = self._forward_part(x_s, label_s, weight_s, x_t_stu, x_t_teas, theta_stu, thetas_tea)
        return self._loss_backward_part(st, gather_activates(st["activates"]))

    def _tail_sums_grads(self):
        return bool(self.fuse_tail and self.sum_grads_in_tail and not _dist_on() and hasattr(self.stu_optimizer, "fused_tail_step")
                    and hasattr(self.student, "pending_grad_sum"))

    def _update(self):
        # Adam, the EMA and the next forwards' weight packs of both networks in ONE sweep when the layout allows ...
        fuse = self.fuse_tail and hasattr(self.stu_optimizer, "fused_tail_step")
        self.fused_last = bool(fuse and self.stu_optimizer.fused_tail_step(self.student, self.teacher, self.tea_optimizer))
        if not self.fused_last:
            if hasattr(self.student, "finish_grads"):
                self.student.finish_grads()     # (a sum left to the fused tail that did not run)
            self.stu_optimizer.step()
            self.tea_optimizer.step()       # EMA after the optimizer step (train_human.py:437-438)


def validate(batches, model, criterion=None):
    """The reference's validate() (train_human.py:461-500) on the device: eval mode, no grad; per-key-point PCK@0.05
    averaged over the set with batch-size weights, entries of -1 (key point absent from a batch) ignored exactly like
    `AverageMeterList(ignore_val=-1)` (lib/meter.py:18-36,65-82), and the batch-size weighted mean loss.  `batches` yields
    (x, label, weight[, meta]).  Decode and PCK run on the device and are ACCUMULATED there: one read-back at the end
    instead of the reference's 2 x 8.4 MB device->host copy + sync per batch.  Returns (acc_per_keypoint list, mean_loss)
    (the caller applies its dataset's group_accuracy)."""
    criterion = criterion or JointsMSELoss()
    was_training = model.training
    model.eval()
    acc_sum = acc_cnt = loss_sum = None
    n_seen = 0
    with torch.no_grad():
        for batch in batches:
            x, label, weight = batch[0], batch[1], batch[2]
            dev = next(model.parameters()).device
            x, label, weight = x.to(dev, non_blocking=True), label.to(dev, non_blocking=True), weight.to(dev, non_blocking=True)
            y = model(x)
            loss = criterion(y, label, weight)
            acc, _, _ = kd.accuracy_device(y, label)
            n = x.shape[0]
            present = (acc != -1).to(torch.float32)
            if acc_sum is None:
                acc_sum, acc_cnt, loss_sum = torch.zeros_like(acc, dtype=torch.float64), torch.zeros_like(acc, dtype=torch.float64), \
                    torch.zeros((), dtype=torch.float64, device=dev)
            acc_sum += (acc * present).double() * n
            acc_cnt += present.double() * n
            loss_sum += loss.detach().double() * n
            n_seen += n
    if was_training:
        model.train()
    if acc_sum is None:
        return [], float("nan")
    avg = torch.where(acc_cnt > 0, acc_sum / acc_cnt.clamp(min=1), torch.zeros_like(acc_sum))     # AverageMeter.avg starts at 0
    out = torch.cat([avg, (loss_sum / max(n_seen, 1)).reshape(1)]).cpu().tolist()                 # the one read-back
    sat = mt.split_saturations(reset=True)          # (the evaluation forward runs the fp32-grade f16x2 mode under 'auto': its range is fp16's)
    if sat:
        import warnings
        warnings.warn(f"validate(): {sat} f16x2 stores saturated at |x| = 65504 (or were NaN) since the last check - activations outside fp16's "
                      "range; run the model with precision='fp32' (exact, slower) to rule the format out")
    return out[:-1], out[-1]


class GraphedTrainStep:
    """The mean-teacher step captured into hipGraphs: ~1300 kernel launches per step are replayed by ONE graph launch on one
    rank (forwards, losses, backward, Adam + EMA + packs), which removes the host launch gaps.
    Data parallel: four graphs - forwards | losses + backward part 1 | backward part 2 | Adam + EMA - with the confidence
    all-gather after the first, the all-reduce of the finished gradient suffix launched after the second (it runs on the
    communicator's stream under the third) and the small prefix all-reduce after the third.
    Inputs are copied into static device tensors before each replay; the re-warp matrices are computed INSIDE the captured step
    (udapose_recon_thetas, double precision) from the batch's raw aug_param values, which the host packs into a pinned ring and
    uploads with one asynchronous copy per step.
    Style transfer (train_human.py:345-358): each direction is its own small graph (content, style, alpha as a device scalar ->
    the step's effective input); the host draws the step's decisions in the reference's order and replays the direction(s) it
    drew, or copies the original images, before the main graph.  Occlusion (train_human.py:374-412): the decisions are taken
    on the device from four uniform draws per sample (trainer.device_occlusion), inside the main graph."""

    def __init__(self, trainer, x_s, label_s, weight_s, x_t_stu, x_t_tea, aug_param_stu, aug_param_tea, warmup=2, split=None, metrics=True,
                 capture_comm=None):
        # metrics: the captured step also decodes y_s and computes PCK@0.05 against label_s on the device (the reference's per-iteration
        # `accuracy(y_s, label_s)`, train_human.py:443) and gathers the losses + PCK into ONE small device vector: step_async() reads
        # it back one step late through a pinned double buffer, so a loop that logs every iteration never drains the device
        self.metrics = bool(metrics)
        self._mvec, self._mpin, self._mev, self._mi, self._mk, self._macc = None, None, [None, None], 0, 0, None
        self.styled = trainer.style_net is not None
        self.occl = trainer.occlude_rate > -1
        if self.occl and not trainer.device_occlusion:
            raise RuntimeError("GraphedTrainStep with occlusion needs trainer.device_occlusion = True (the reference's host draws read "
                               "the confidences back every step, which a captured step cannot do)")
        self.t = trainer
        self._stage, self._have_staged = None, False
        dev = x_s.device
        n = x_s.shape[0]
        self.n = n
        # The re-warp matrices are computed INSIDE the captured step (udapose_recon_thetas, double precision) from the batch's raw
        # aug_param values: per step the host packs 2 x N x 6 doubles into a pinned buffer and issues one asynchronous copy.
        self.static = {"x_s": x_s.clone(), "label_s": label_s.clone(), "weight_s": weight_s.clone(), "x_t_stu": x_t_stu.clone(),
                       "x_t_tea": x_t_tea.clone(),
                       "aug": torch.empty(2, n, 6, dtype=torch.float64, device=dev),
                       "theta_stu": torch.empty(n, 3, 6, dtype=torch.float32, device=dev),
                       "theta_tea": torch.empty(n, 3, 6, dtype=torch.float32, device=dev)}
        st = self.static
        self._aug_pin = [torch.empty(2, n, 6, dtype=torch.float64).pin_memory() for _ in range(4)]
        self._aug_ev = [None] * 4
        self._aug_i = 0
        self._aug_last = None
        self._stage_aug(aug_param_stu, aug_param_tea)
        # the main graph reads the step's EFFECTIVE inputs: the originals, or what the style graphs wrote
        st["x_s_in"] = st["x_s"].clone() if self.styled else st["x_s"]
        st["x_t_tea_in"] = st["x_t_tea"].clone() if self.styled else st["x_t_tea"]
        if self.styled:
            st["alpha_s2t"] = torch.ones(1, dtype=torch.float32, device=dev)
            st["alpha_t2s"] = torch.ones(1, dtype=torch.float32, device=dev)
        if self.occl:
            st["theta_back"] = torch.empty(n, 1, 6, dtype=torch.float32, device=dev)
            st["u"] = torch.zeros(n, 4, dtype=torch.float32, device=dev)
            trainer._occl = ("device", st["theta_back"], st["u"])
        else:
            trainer._occl = None
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        self.g_style = {}
        self._cap = {"stream": torch.cuda.Stream(device=dev, priority=trainer.stream_priority)} if trainer.stream_priority else {}
        with torch.cuda.stream(side):
            if self.styled:             # (the style net's plans and packs; no model state involved)
                self._style_pass("enc")
                self._style_pass("s2t")
                self._style_pass("t2s")
            for _ in range(warmup):     # REAL steps (same host draws as step()): fill the plan / table caches, allocator steady state
                self._thetas()
                self._draw_and_style()
                trainer._forward_backward(st["x_s_in"], st["label_s"], st["weight_s"], st["x_t_stu"], [st["x_t_tea_in"]], st["theta_stu"],
                                          [st["theta_tea"]])
                trainer._sync_grads()
                trainer._update()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if _dist_on() and dist.get_backend() == "nccl":
            # RCCL's watchdog thread polls the end events of the EAGER collectives issued so far (the warm-up steps'); on this HIP an
            # event query fails with hipErrorCapturedEvent once the communicator's stream has joined a capture, even for an event that
            # was recorded before it - and the watchdog then takes the process down.  Everything is complete here (synchronize above):
            # give the watchdog (100 ms poll) time to retire those works before any collective is captured.  torch exposes no call that
            # waits for the watchdog's list to drain, so this IS a timed wait (ten polls) - which is why capturing the
            # collectives is opt-in (capture_comm below) and why the capture's failure path falls back to eager collectives.
            if capture_comm or (capture_comm is None and os.environ.get("UDAPOSE_CAPTURE_COMM", "0") == "1"):
                import time
                dist.barrier()
                torch.cuda.synchronize()
                time.sleep(1.0)
        style_mode = "thread_local" if _dist_on() else "global"
        if self.styled:
            for which in ("enc", "s2t", "t2s"):
                g = torch.cuda.CUDAGraph()
                pool = {} if not self.g_style else {"pool": self.g_style["enc"].pool()}
                with torch.cuda.graph(g, capture_error_mode=style_mode, **pool):
                    self._style_pass(which)
                self.g_style[which] = g
        # Data parallel: the confidence all-gather sits between the forwards and the losses, the gradient all-reduce between
        # backward and the optimizer; both stay eager, so the step is cut into three graphs around them.
        # capture_comm (round 5): the step's collectives - confidence all-gather, gradient all-reduce buckets - are captured INTO the step's
        # graph (RCCL launches are stream-ordered kernels: capturable), so the data-parallel step is ONE graph launch like the one-rank step
        # instead of four graphs with eager collectives between them.  OPT-IN (round 6, ADVICE r5: the argument, or UDAPOSE_CAPTURE_COMM=1):
        # it rests on a timed wait for RCCL's watchdog (above), replayed collectives are invisible to torch.distributed's timeout, and no
        # multi-GPU node has run it yet; bench.py --gpus N asks for it explicitly (with its own hang guard) and lets the ranks fall back
        # together.  If the capture raises (a collective backend that cannot be captured), the four-graph form below is built instead.
        if capture_comm is None:
            capture_comm = (_dist_on() and dist.get_backend() == "nccl" and split is None
                            and os.environ.get("UDAPOSE_CAPTURE_COMM", "0") == "1")
        self.capture_comm = bool(capture_comm) and _dist_on()
        if self.capture_comm and trainer.sync.comm_dtype == "bf16" and trainer._overlap():
            # the bf16 exchange of the suffix runs on a side stream that hands over to the communicator's stream and waits for it again:
            # a fork that waits back on a non-origin stream brings hipStreamEndCapture down on this ROCm (tools/capture_fork_patterns.py)
            self.capture_comm = False
        trainer.capture_comm = self.capture_comm
        self.split = (_dist_on() and not self.capture_comm) if split is None else bool(split)
        # hyper-parameters that are kernel ARGUMENTS of the captured launches stay what they were at capture: step() checks
        # them.  (lr / grad_scale are read from device memory and follow the optimizer's param_groups, see optim.py.)
        self._frozen = self._frozen_hyper()
        token = object()
        for m in (trainer.student, trainer.teacher):
            m._capture_token = token            # weight packs: captured once per network, unconditionally (pose_resnet._pack)
        self.g_fb, self.g_up = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        # other threads (the RCCL watchdog of torch.distributed) may touch the runtime while this thread captures
        mode = "thread_local" if _dist_on() else "global"
        # overlap: the backward is cut after layer3 (two graph segments) and the finished gradient suffix is all-reduced on
        # the communicator's stream while the second segment replays
        self.overlap = trainer._overlap()
        self.g_lb2 = None
        # one rank, nothing eager between backward and the optimizer: the update is captured into the same graph (one launch per step)
        self.one_graph = (not self.split) and (self.capture_comm or not _dist_on()) and trainer.single_graph
        self.capture_fallback = None        # why the captured-collectives form was given up (None: it was not, or was never asked for)
        if not self.split and self.capture_comm:
            e = None
            try:
                self._capture_one(trainer, st, mode)
            except Exception as e_:      # (the backend refused the capture: fall back to eager collectives between four graphs)
                e = e_
            # The fallback is a COLLECTIVE decision: a rank that kept its captured graph while another replays eager collectives would pair
            # them wrongly and hang.  Every rank reports, the minimum decides (one eager all-reduce, outside any capture).
            ok_here = e is None
            if dist.is_available() and dist.is_initialized() and not torch.cuda.is_current_stream_capturing():
                flag = torch.tensor([1.0 if ok_here else 0.0], device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                ok_all = bool(flag.item() >= 1.0)
            else:
                ok_all = ok_here
            if not ok_all:
                import warnings
                why = f"{type(e).__name__}: {e}" if e is not None else "another rank could not capture the collectives"
                self.capture_fallback = why
                warnings.warn(f"GraphedTrainStep: capturing the collectives into the step's graph failed ({why}); "
                              "every rank uses the four-graph form with eager collectives")
                trainer.capture_comm = False
                self.capture_comm = False
                trainer._metrics_cb = None
                self._macc = None
                trainer.student._pending_lower, trainer.student._pending_wg = [], []
                # (whatever the aborted capture left half-done in the gradient exchange: the Work of the suffix's collective, its view,
                # the bf16 exchange's side stream - the four-graph form starts from a clean GradSync)
                trainer.sync._work, trainer.sync._upper, trainer.sync._cstream = None, None, None
                trainer.student._grad_state = None
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("GraphedTrainStep: the failed capture of the collectives left the stream capturing; cannot fall back")
                torch.cuda.synchronize()
                self.g_fb = torch.cuda.CUDAGraph()
                self.split = True
                self.one_graph = False
        if not self.split and self.capture_comm:
            pass
        elif not self.split:
            if self.one_graph and self.metrics:
                trainer._metrics_cb = self._metrics_side
            try:
                with torch.cuda.graph(self.g_fb, capture_error_mode=mode, **self._cap):
                    self._thetas()
                    self.out = trainer._forward_backward(st["x_s_in"], st["label_s"], st["weight_s"], st["x_t_stu"], [st["x_t_tea_in"]],
                                                         st["theta_stu"], [st["theta_tea"]])
                    if trainer.student._pending_lower:      # (overlap forced on one rank: both backward parts in the one graph)
                        trainer._backward_lower()
                    if self.one_graph:
                        trainer._sync_grads()
                        trainer._update()
                        self._capture_metrics()
            finally:
                trainer._metrics_cb = None
        else:
            self.g_lb = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_fb, capture_error_mode=mode, **self._cap):
                self._thetas()
                self.fwd_state = trainer._forward_part(st["x_s_in"], st["label_s"], st["weight_s"], st["x_t_stu"], [st["x_t_tea_in"]],
                                                       st["theta_stu"], [st["theta_tea"]])
            g0 = gather_activates(self.fwd_state["activates"])
            self.gathered = g0.clone() if g0 is not None else self.fwd_state["activates"].reshape(-1).clone()
            with torch.cuda.graph(self.g_lb, pool=self.g_fb.pool(), capture_error_mode=mode, **self._cap):
                self.out = trainer._loss_backward_part(self.fwd_state, self.gathered)
            if self.overlap:
                trainer.sync.start_upper()
                self.g_lb2 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.g_lb2, pool=self.g_fb.pool(), capture_error_mode=mode, **self._cap):
                    trainer._backward_lower()
        if self.one_graph:
            self.g_up = None
        else:
            if self.capture_comm:
                pass                        # (the collectives are inside g_fb)
            elif self.g_lb2 is not None:
                trainer.sync.finish()
            else:
                trainer.sync()
            with torch.cuda.graph(self.g_up, pool=self.g_fb.pool(), capture_error_mode=mode, **self._cap):
                trainer._update()
                self._capture_metrics()
        for m in (trainer.student, trainer.teacher):
            m._capture_token = None
            m.weights_changed()
        self._opt_key = self._optimizer_key()
        self._fused_tail = bool(trainer.fused_last)
        self._maintained = [(trainer.student, trainer.student._last_hd, True), (trainer.teacher, trainer.teacher._last_hd, False)]
        if self._fused_tail:
            for m, hd, bwd in self._maintained:
                m.packs_refreshed(hd, bwd)
        torch.cuda.synchronize()

    def release(self):
        """Destroy the captured graphs.  With capture_comm the graphs hold RCCL launches, and RCCL's communicator teardown waits for every
        such graph to be gone: call this (or drop every reference to the object) BEFORE torch.distributed.destroy_process_group(), which
        otherwise never returns (measured on RCCL 2.26 / ROCm 7.0)."""
        self.g_fb = self.g_lb = self.g_lb2 = self.g_up = None
        self.g_style = {}
        import gc
        gc.collect()
        torch.cuda.synchronize()

    def __del__(self):
        # (graphs that captured RCCL launches must be gone before destroy_process_group(): dropping the last reference does what release() does)
        try:
            if getattr(self, "capture_comm", False):
                self.g_fb = self.g_lb = self.g_lb2 = self.g_up = None
        except Exception:
            pass

    def _capture_one(self, trainer, st, mode):
        """The whole data-parallel step - forwards, all-gather, losses, backward part 1, the suffix's all-reduce under backward part 2, the
        prefix's all-reduce, Adam + EMA + packs - captured into self.g_fb."""
        if os.environ.get("UDAPOSE_TEST_FAIL_CAPTURE", "0") == "1":       # test hook (tests/test_gpu_hotpath.py): exercises the collective fallback
            raise RuntimeError("injected capture failure (UDAPOSE_TEST_FAIL_CAPTURE=1)")
        if self.one_graph and self.metrics:
            trainer._metrics_cb = self._metrics_side
        try:
            with torch.cuda.graph(self.g_fb, capture_error_mode=mode, **self._cap):
                self._thetas()
                self.out = trainer._forward_backward(st["x_s_in"], st["label_s"], st["weight_s"], st["x_t_stu"], [st["x_t_tea_in"]],
                                                     st["theta_stu"], [st["theta_tea"]])
                trainer._sync_grads()
                if self.one_graph:
                    trainer._update()
                    self._capture_metrics()
        finally:
            trainer._metrics_cb = None

    def _capture_metrics(self):
        """(inside the capture of the step's last graph) losses + device PCK of the source batch -> self._mvec = [loss_all, loss_s, loss_c,
        avg_acc, cnt, acc[0..K-1]]: decode and PCK are the udapose_heatmap_argmax / udapose_pck launches of lib.keypoint_detection."""
        if not self.metrics:
            return
        if self._macc is not None:          # computed on the side stream while the backward ran: join it
            torch.cuda.current_stream().wait_stream(self._macc[2])
            acc, avg_cnt = self._macc[0], self._macc[1]
            for t in (acc, avg_cnt):
                t.record_stream(torch.cuda.current_stream())
        else:
            acc, avg_cnt, _ = kd.accuracy_device(self.out["y_s"], self.static["label_s"])
        parts = [self.out["loss_all"].reshape(1), self.out["loss_s"].reshape(1), self.out["loss_c"].reshape(1), avg_cnt.reshape(2), acc.reshape(-1)]
        self._mk = int(acc.numel())
        self._mvec = torch.cat([p.float() for p in parts])
        self.out["acc_s"], self.out["acc_avg_cnt"] = acc, avg_cnt

    def _metrics_side(self, st):
        """(inside the one-graph capture, called by the trainer before the losses) the metric's decode + PCK launches on the teacher's
        stream, which is idle from here on: they run under the backward instead of behind the optimizer tail (pck_k alone is a 24 us
        single-work-group kernel)."""
        t = self.t
        side = t._side[0] if (t._side is not None and t.concurrent) else None
        if side is None:
            return
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            acc, avg_cnt, _ = kd.accuracy_device(st["y_s"], self.static["label_s"])
        st["y_s"].record_stream(side)
        self._macc = (acc, avg_cnt, side)

    def _metrics_dict(self, v):
        v = v.tolist()
        return {"loss_all": v[0], "loss_s": v[1], "loss_c": v[2], "acc_s": v[3], "cnt_s": int(v[4]), "acc_per_keypoint": v[5:5 + self._mk]}

    def step_async(self, *args, **kw):
        """step() with the metric read-back DEFERRED by one step: replays this step, queues the copy of its metric vector into a pinned
        slot behind it, and only then waits for the PREVIOUS step's slot - the device always has the next replay queued while the host
        reads, so the reference's per-iteration logging (train_human.py:440-452) costs no device idle time (a synchronous read exposes
        the ~0.75 ms launch latency of a 1270-node graph every step).  Returns the previous step's metrics as Python numbers (None on
        the first call); flush_metrics() returns the last step's."""
        if self._mvec is None:
            raise RuntimeError("GraphedTrainStep(metrics=False) has no metric vector to read back")
        if self._mpin is None:
            self._mpin = [torch.empty(self._mvec.numel(), dtype=torch.float32).pin_memory() for _ in range(2)]
        self.step(*args, **kw)
        i = self._mi
        self._mpin[i].copy_(self._mvec, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._mev[i] = ev
        self._mi = 1 - i
        prev = self._mev[1 - i]
        if prev is None:
            return None
        prev.synchronize()
        self._mev[1 - i] = None
        return self._metrics_dict(self._mpin[1 - i])

    def flush_metrics(self):
        """Metrics of the most recent step_async() (waits for it)."""
        last = 1 - self._mi
        ev = self._mev[last]
        if ev is None:
            return None
        ev.synchronize()
        self._mev[last] = None
        return self._metrics_dict(self._mpin[last])

    profile_segments = False        # bench.py --dp-segments: HIP events between the parts of a step (device time per part, after a synchronize)

    def _seg_mark(self, name):
        if not self.profile_segments:
            return
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self.__dict__.setdefault("_seg_events", []).append((name, e))

    def segment_ms(self):
        """Mean device time of each part of the step since the last call: {part: ms} (the parts of one step are consecutive events)."""
        ev = self.__dict__.pop("_seg_events", [])
        torch.cuda.synchronize()
        tot, cnt = {}, {}
        for (n0, e0), (n1, e1) in zip(ev, ev[1:]):
            if n1 == "start":
                continue
            tot[n1] = tot.get(n1, 0.0) + e0.elapsed_time(e1)
            cnt[n1] = cnt.get(n1, 0) + 1
        return {k: round(tot[k] / cnt[k], 3) for k in tot}

    def _stage_aug(self, aug_param_stu=None, aug_param_tea=None):
        """The batch's raw aug_param values -> the static [2,N,6] float64 device buffer the captured udapose_recon_thetas launches
        read (a ring of pinned host buffers and one asynchronous copy: nothing here waits for the device)."""
        if aug_param_stu is None and aug_param_tea is None:
            return
        i = self._aug_i = (self._aug_i + 1) % len(self._aug_pin)
        if self._aug_ev[i] is not None:
            self._aug_ev[i].synchronize()             # (the copy issued from this slot four steps ago: long done)
        pin = self._aug_pin[i]
        if self._aug_last is not None:
            pin.copy_(self._aug_last)                 # a side that is not given keeps its previous values
        if aug_param_stu is not None:
            warp.pack_aug_param(aug_param_stu, self.n, out=pin[0])
        if aug_param_tea is not None:
            warp.pack_aug_param(aug_param_tea, self.n, out=pin[1])
        self.static["aug"].copy_(pin, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._aug_ev[i], self._aug_last = ev, pin

    def _thetas(self):
        """Launch the matrix kernels (captured: at the head of the step's first graph) from the static aug buffer.  (Round 4 tried them on a
        stream of their own, beside the forwards: no measurable gain, and the earlier start of the three branches changed the arrival order
        of the stem's fp32 atomics often enough to break the bit-for-bit twins of tests/test_gpu_steps.py; reverted.)"""
        st, r = self.static, self.t.ratio
        warp.thetas_from_packed(st["aug"][0], r, want_fwd=True, want_back=self.occl, fwd=st["theta_stu"], back=st.get("theta_back"))
        warp.thetas_from_packed(st["aug"][1], r, fwd=st["theta_tea"])

    def _style_pass(self, which):
        """"enc": relu4_1 of both ORIGINAL batches into static feature buffers (each image is encoded once per step, whichever
        directions are drawn); "s2t" / "t2s": AdaIN + decoder of one direction from those features into the main graph's input buffer."""
        st, t = self.static, self.t
        net = getattr(t.style_net, "module", t.style_net)
        with torch.no_grad():
            if which == "enc":
                f_s, f_t = net.encode_features(st["x_s"]), net.encode_features(st["x_t_tea"])
                if "f_s" not in st:
                    st["f_s"], st["f_t"] = torch.empty_like(f_s), torch.empty_like(f_t)
                st["f_s"].copy_(f_s)
                st["f_t"].copy_(f_t)
            elif which == "s2t":
                st["x_s_in"].copy_(net.transfer_from_features(st["f_s"], st["f_t"], st["alpha_s2t"], clamp=t.recover))
            else:
                st["x_t_tea_in"].copy_(net.transfer_from_features(st["f_t"], st["f_s"], st["alpha_t2s"], clamp=t.recover))

    def _draw_and_style(self):
        """The step's host draws, in the eager step's (= the reference's) order, and what follows from them: each style
        direction drawn runs (its graph once captured) into the main graph's input buffer, the other keeps the original
        images; the occlusion's uniform numbers go to the device."""
        st = self.static
        if self.styled:
            decisions = self.t.draw_style_decisions()
            if any(a is not None for a in decisions):
                if "enc" in self.g_style:
                    self.g_style["enc"].replay()
                else:
                    self._style_pass("enc")
            for which, a, src in zip(("s2t", "t2s"), decisions, (("x_s_in", "x_s"), ("x_t_tea_in", "x_t_tea"))):
                if a is not None:
                    st["alpha_" + which].fill_(float(a))
                    if which in self.g_style:
                        self.g_style[which].replay()
                    else:
                        self._style_pass(which)
                else:
                    st[src[0]].copy_(st[src[1]], non_blocking=True)
        if self.occl:
            st["u"].copy_(self.t.draw_occlusion_uniforms(self.n), non_blocking=True)

    def _optimizer_key(self):
        """Storage the captured optimizer launches point at: the device state of every group and the first moment tensor."""
        opt = self.t.stu_optimizer
        key = [ent[0].data_ptr() for _, ent in sorted(getattr(opt, "_dev", {}).items())]
        for g in opt.param_groups:
            for p in g["params"]:
                st = opt.state.get(p)
                if st:
                    key += [v.data_ptr() for v in st.values() if torch.is_tensor(v)]
                    break
        return key

    def _frozen_hyper(self):
        t = self.t
        opt = t.stu_optimizer
        frozen = [("teacher_alpha", float(t.tea_optimizer.alpha)), ("lambda_c", float(t.lambda_c)), ("mask_ratio", float(t.mask_ratio)),
                  ("sigma", float(t.sigma)), ("bn_momentum", float(t.student.bn_momentum)), ("occlude_rate", float(t.occlude_rate)),
                  ("occlude_thresh", float(t.occlude_thresh)), ("occlude_size", int(t.occlude_size))]
        for gi, g in enumerate(opt.param_groups):
            for k in ("betas", "eps", "weight_decay", "momentum", "nesterov"):
                if k in g:
                    frozen.append((f"group{gi}.{k}", g[k]))
        return frozen

    def prefetch(self, x_s, label_s, weight_s, x_t_stu, x_t_tea):
        """Start copying the NEXT batch (pinned host tensors) into staging buffers on a copy stream; it overlaps with the
        replay of the current step.  The following step() call (with no tensors given) takes the staged batch with five
        device-to-device copies (92 MB, ~30 us) instead of waiting for PCIe on the compute stream."""
        if self._stage is None:
            self._stage = {k: torch.empty_like(self.static[k]) for k in ("x_s", "label_s", "weight_s", "x_t_stu", "x_t_tea")}
            self._copy_stream = torch.cuda.Stream(device=self.static["x_s"].device)
            self._staged, self._consumed = torch.cuda.Event(), None
        if self._consumed is not None:
            self._copy_stream.wait_event(self._consumed)     # the previous step's D2D reads of the staging buffers come first
        with torch.cuda.stream(self._copy_stream):
            for k, v in (("x_s", x_s), ("label_s", label_s), ("weight_s", weight_s), ("x_t_stu", x_t_stu), ("x_t_tea", x_t_tea)):
                self._stage[k].copy_(v, non_blocking=True)
            self._staged.record(self._copy_stream)
        self._have_staged = True

    def step(self, x_s=None, label_s=None, weight_s=None, x_t_stu=None, x_t_tea=None, aug_param_stu=None, aug_param_tea=None):
        st = self.static
        if x_s is None and self._have_staged:
            torch.cuda.current_stream().wait_event(self._staged)
            for k in self._stage:
                st[k].copy_(self._stage[k], non_blocking=True)
            self._consumed = torch.cuda.Event()
            self._consumed.record()
            self._have_staged = False
        for k, v in (("x_s", x_s), ("label_s", label_s), ("weight_s", weight_s), ("x_t_stu", x_t_stu), ("x_t_tea", x_t_tea)):
            if v is not None and v.data_ptr() != st[k].data_ptr():
                st[k].copy_(v, non_blocking=True)
        self._stage_aug(aug_param_stu, aug_param_tea)
        frozen_now = self._frozen_hyper()
        if frozen_now != self._frozen:
            changed = [a[0] for a, b in zip(frozen_now, self._frozen) if a != b]
            raise RuntimeError(f"GraphedTrainStep: {changed} changed after capture; these are baked into the captured launches - "
                               "build a new GraphedTrainStep (lr and grad_scale may change freely)")
        if self._optimizer_key() != self._opt_key:
            raise RuntimeError("GraphedTrainStep: the optimizer's state tensors were replaced after capture (a new optimizer, or state moved to "
                               "another device): the captured launches still point at the old storage - build a new GraphedTrainStep "
                               "(FusedAdam / FusedSGD.load_state_dict restores IN PLACE and is fine)")
        if self._fused_tail:
            # the captured step relies on the packs its own previous update left: if anything else touched the weights since
            # (an eager optimizer, load_state_dict), re-pack eagerly first
            for m, hd, bwd in self._maintained:
                if hd.wpack_version is None or hd.wpack_version[0] != m.version_key():
                    hd.wpack_version = None
                    with torch.enable_grad() if bwd else torch.no_grad():
                        m.prepare(self.static["x_s"])
        self._draw_and_style()
        seg = self._seg_mark            # (profile_segments: an event after each part of the data-parallel step; a no-op otherwise)
        seg("start")
        if self.one_graph:
            self.t.stu_optimizer.sync_hyper()    # lr scheduler / loss scale -> device state read by the captured sweep
        self.g_fb.replay()
        seg("forwards")
        if self.split:
            g = gather_activates(self.fwd_state["activates"])
            self.gathered.copy_(g if g is not None else self.fwd_state["activates"].reshape(-1))
            seg("gather")
            self.g_lb.replay()
            seg("losses+backward1")
        if self.g_lb2 is not None:
            self.t.sync.start_upper()            # suffix (94 %) on the communicator's stream ...
            self.g_lb2.replay()                  # ... under backward part 2
            seg("backward2")
            self.t.sync.finish()
            seg("finish")
        elif not self.one_graph and not self.capture_comm:
            self.t.sync()
            seg("allreduce")
        if not self.one_graph:
            self.t.stu_optimizer.sync_hyper()    # lr scheduler / loss scale -> device state read by the captured sweep
            self.g_up.replay()
            seg("update")
        # the replayed Adam / EMA kernels changed both networks' parameters behind torch's back: every executor plan (other
        # batch sizes, validate(), fp32 mode) must re-pack its bf16 weights before its next forward
        self.t.student.weights_changed()
        self.t.teacher.weights_changed()
        if self._fused_tail:
            for m, hd, bwd in self._maintained:          # ... whose own packs the captured update has just rewritten
                m.packs_refreshed(hd, bwd)
        return self.out
