"""Fused multi-tensor optimizers on MI355X kernels: drop-in torch.optim.Optimizer subclasses for the reference's
`Adam(student.parameters(), lr)` / `SGD(..., momentum=0.9, weight_decay=1e-4, nesterov=True)` (train_human.py:136-139).
One kernel launch sweeps all parameter tensors (28 B/param for Adam), instead of torch's per-group foreach chains.

Graph-replay safe: the step counter, the learning rate and the gradient scale live in an 8-float device tensor per
parameter group (`udapose_adam_multi`'s dev_state).  `sync_hyper()` uploads lr / grad_scale when the host values changed
(an lr scheduler, a loss-scale update); GraphedTrainStep calls it before every replay.  `state_dict()` is torch's plain
layout (device tables and state tensors are kept on the optimizer object, not in param_groups) with `step` read back from
the device counter; checkpoints written by torch.optim.Adam / SGD load (per-parameter `step` entries are folded into the
group counter).

Loss scaling for the fp16 precision (`dynamic_loss_scale=True`) is torch.cuda.amp.GradScaler's algorithm
(train_human.py:260,285-287) kept entirely on the device: `scale_loss(loss)` multiplies by the device-resident scale,
`step()` = found-inf sweep over the gradients -> Adam / SGD sweep that un-scales (and is skipped whole, counter included,
when an inf / nan was found) -> scale update (x backoff on overflow, x growth after `growth_interval` clean steps).
No host read-back, so the whole thing is hipGraph-capturable.
"""
import torch

from . import _hip
from ._hip import check, lib, ptr
from .utils import _MultiTensorTable, _bump_versions


class _FusedBase(torch.optim.Optimizer):
    _state_names = ()

    def __init__(self, params, defaults, scaler=None):
        super().__init__(params, defaults)
        self._tables = {}        # group index -> _MultiTensorTable (raw device pointers: never serialised)
        self._dev = {}           # group index -> (device state tensor [8], uploaded (lr, grad_scale))
        # dynamic loss scaling (GradScaler's defaults: 65536, x2 every 2000 clean steps, x0.5 on overflow), or None
        self._scaler = scaler
        if scaler is not None and len(self.param_groups) != 1:
            raise ValueError("dynamic loss scaling needs a single parameter group (one device-resident scaler state)")

    # ------------------------------------------------------------------ tables / device state
    def _gather(self, gi, group):
        ps = [p for p in group["params"] if p.grad is not None]
        if not ps:
            return None
        for p in ps:
            _hip.require_cuda(p)
            if p.dtype != torch.float32 or p.grad.dtype != torch.float32:
                raise RuntimeError("fused optimizers need fp32 parameters and gradients")
            if p.grad.stride() != p.stride():
                p.grad = p.grad.clone(memory_format=torch.preserve_format).as_strided(p.shape, p.stride()).copy_(p.grad)
            st = self.state[p]
            for n in self._state_names:
                if n not in st or not torch.is_tensor(st[n]) or st[n].device != p.device or st[n].stride() != p.stride():
                    old = st.get(n)
                    st[n] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    if torch.is_tensor(old) and old.shape == p.shape:     # state loaded from a checkpoint (CPU / other layout)
                        st[n].copy_(old)
        lists = [[p.data for p in ps], [p.grad for p in ps]] + [[self.state[p][n] for p in ps] for n in self._state_names]
        key = _MultiTensorTable.key_of(lists)
        tab = self._tables.get(gi)
        if tab is None or tab.key != key:
            tab = _MultiTensorTable(lists)
            self._tables[gi] = tab
        return ps, tab

    def _dev_state(self, gi, group, device):
        """[step, bc1, bc2sqrt, lr, grad_scale, 0, 0, 0] on the device; created from the host's group values."""
        ent = self._dev.get(gi)
        if ent is None or ent[0].device != device:
            host, hyper = self._host_state(group)
            ent = [torch.tensor(host, dtype=torch.float32, device=device), hyper]
            self._dev[gi] = ent
        return ent

    def _host_state(self, group):
        """The 8 floats of a group's device state from the host's group values, and the (lr, grad_scale) pair they carry."""
        hyper = (float(group["lr"]), float(group.get("grad_scale", 1.0)))
        scale, tracker = 0.0, 0.0
        if self._scaler is not None:
            scale = float(group.get("loss_scale", self._scaler["init_scale"]))
            tracker = float(group.get("growth_tracker", 0))
            hyper = (hyper[0], 1.0 / scale)
        return [float(group.get("step", 0)), 0.0, 0.0, hyper[0], hyper[1], 0.0, scale, tracker], hyper

    # ------------------------------------------------------------------ loss scaling (fp16)
    def loss_scale(self):
        """The current loss scale as a 0-d DEVICE tensor (1.0 without dynamic scaling); no host synchronisation."""
        if self._scaler is None:
            return None
        p0 = self.param_groups[0]["params"][0]
        return self._dev_state(0, self.param_groups[0], p0.device)[0][6]

    def scale_loss(self, loss):
        """loss * scale (GradScaler.scale); identity without dynamic scaling."""
        sc = self.loss_scale()
        return loss if sc is None else loss * sc

    def _pre_sweep(self, t, ent, grad2_delta=0):
        """found-inf sweep of the loss scaler; grad2_delta: byte distance to a second per-pass gradient buffer whose sum is still pending
        (the fused tail adds it itself): the check then looks at g + g2."""
        if self._scaler is not None:
            check(lib().udapose_grad_scaler_check2(_hip.stream(), ptr(t.ptrs[1]), ptr(t.sizes), ptr(t.blk_t), ptr(t.blk_o), t.nblocks, ptr(ent[0]),
                                                   int(grad2_delta)), "grad_scaler_check")

    def _post_sweep(self, ent):
        if self._scaler is not None:
            sc = self._scaler
            check(lib().udapose_grad_scaler_update(_hip.stream(), ptr(ent[0]), float(sc["growth_factor"]), float(sc["backoff_factor"]),
                                                   int(sc["growth_interval"])), "grad_scaler_update")

    def sync_hyper(self):
        """Upload lr / grad_scale of every group whose host value changed since the last upload (stream-ordered 8-byte copy;
        call before replaying a captured step)."""
        for gi, group in enumerate(self.param_groups):
            ent = self._dev.get(gi)
            if ent is None:
                continue
            if self._scaler is not None:           # grad_scale = 1 / loss scale is owned by the device-side scaler
                if float(group["lr"]) != ent[1][0]:
                    ent[0][3:4].copy_(torch.tensor([float(group["lr"])], dtype=torch.float32), non_blocking=True)
                    ent[1] = (float(group["lr"]), ent[1][1])
                continue
            hyper = (float(group["lr"]), float(group.get("grad_scale", 1.0)))
            if hyper != ent[1]:
                ent[0][3:5].copy_(torch.tensor(hyper, dtype=torch.float32), non_blocking=True)
                ent[1] = hyper

    # ------------------------------------------------------------------ checkpoints
    def state_dict(self):
        for gi, group in enumerate(self.param_groups):
            ent = self._dev.get(gi)
            if ent is not None:
                host = ent[0].tolist()
                group["step"] = int(round(host[0]))                       # replays tick the device counter only
                if self._scaler is not None:
                    group["loss_scale"], group["growth_tracker"] = float(host[6]), int(round(host[7]))
        return super().state_dict()

    def load_state_dict(self, state_dict):
        """torch's load, then (1) the group step counter from wherever the checkpoint keeps it, (2) IN PLACE: moment / momentum tensors
        and device state that already exist keep their storage and receive the loaded values, so kernels captured in a hipGraph
        (GraphedTrainStep), which hold raw pointers to them, go on working on the restored state."""
        old_state = {p: dict(st) for p, st in self.state.items()}
        old_dev = dict(self._dev)
        super().load_state_dict(state_dict)
        for group in self.param_groups:
            if "step" not in group or group["step"] is None:
                # torch.optim.Adam / SGD checkpoints keep `step` per parameter (or not at all)
                steps = [float(self.state[p]["step"]) for p in group["params"] if p in self.state and "step" in self.state[p]]
                group["step"] = int(max(steps)) if steps else 0
                if not steps and any(torch.is_tensor(self.state.get(p, {}).get("momentum_buffer")) for p in group["params"]):
                    # torch.optim.SGD keeps no step at all: restored momentum buffers mean "not the first step" (the first step
                    # INITIALISES the buffer with the gradient, train_human.py:136,157,231 resume SGD runs from such checkpoints)
                    group["step"] = 1
            group.setdefault("grad_scale", 1.0)
        for p, st in self.state.items():
            o = old_state.get(p)
            if not o:
                continue
            for n in self._state_names:
                a, b = o.get(n), st.get(n)
                if torch.is_tensor(a) and torch.is_tensor(b) and a.shape == b.shape and a.device == p.device and a.stride() == p.stride():
                    a.copy_(b)
                    st[n] = a
        self._dev = {}
        for gi, group in enumerate(self.param_groups):
            ent = old_dev.get(gi)
            if ent is not None:
                host, hyper = self._host_state(group)
                ent[0].copy_(torch.tensor(host, dtype=torch.float32))
                ent[1] = hyper
                self._dev[gi] = ent
        # (job tables are keyed by the tensors' pointers: they stay valid where the storage was kept, and are rebuilt otherwise)


class FusedAdam(_FusedBase):
    _state_names = ("exp_avg", "exp_avg_sq")

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, grad_scale=1.0, dynamic_loss_scale=False,
                 init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000):
        scaler = dict(init_scale=init_scale, growth_factor=growth_factor, backoff_factor=backoff_factor,
                      growth_interval=growth_interval) if dynamic_loss_scale else None
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, grad_scale=grad_scale, step=0), scaler)

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        capturing = torch.cuda.is_current_stream_capturing()
        for gi, group in enumerate(self.param_groups):
            got = self._gather(gi, group)
            if got is None:
                continue
            ps, t = got
            ent = self._dev_state(gi, group, ps[0].device)          # (created with the number of steps done so far)
            group["step"] = group.get("step", 0) + 1
            b1, b2 = group["betas"]
            if not capturing:
                self.sync_hyper()
            self._pre_sweep(t, ent)
            check(lib().udapose_adam_multi(_hip.stream(), ptr(t.ptrs[0]), ptr(t.ptrs[1]), ptr(t.ptrs[2]), ptr(t.ptrs[3]), ptr(t.sizes), ptr(t.blk_t),
                                           ptr(t.blk_o), t.nblocks, float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                           float(group["weight_decay"]), int(group["step"]), float(group.get("grad_scale", 1.0)), ptr(ent[0])),
                  "adam_multi")
            self._post_sweep(ent)
            _bump_versions(ps)
        return loss


    # ------------------------------------------------------------------ fused tail: Adam + EMA + weight packs in one sweep
    def fused_tail_step(self, student, teacher, ema):
        """One launch for torch.optim.Adam.step on the student, OldWeightEMA.step into the teacher and the weight packs of the
        two executor plans their last forwards used (udapose_net_fused_update).  Returns False - nothing done - when the
        layout is not the one the kernel serves (then step() and ema.step() apply)."""
        import ctypes as C
        if len(self.param_groups) != 1:
            return False
        group = self.param_groups[0]
        ps = list(student.parameters())
        hd_s, hd_t = getattr(student, "_last_hd", None), getattr(teacher, "_last_hd", None)
        # the teacher either shares the student's 16-bit element type (its packs are written by the sweep) or runs the fp32-grade
        # f16x2 mode (the reference's precision mix: its split packs are refreshed by one pack launch right after the sweep)
        if hd_s is None or hd_t is None or hd_s.precision not in ("bf16", "fp16"):
            return False
        t_split = hd_t.precision == "f16x2"
        # both plans are handed to ONE library (the student's): the teacher's must have been created by it (PoseResNet.aux_lib_kind puts an
        # f16x2 teacher's plan into the student's build; a plan of the other .so is never dereferenced here - ADVICE r3)
        if hd_s.L is not hd_t.L and t_split and hasattr(teacher, "aux_lib_kind") and teacher.aux_lib_kind != hd_s.precision:
            # an 'auto' student resolved to the other build than the one the teacher's f16x2 plan was created in (the engine guesses the
            # build before the first forward): move the teacher's fp32-grade plans into the student's build - the next teacher forward
            # creates its plan there, and the one-launch tail applies from the following step on
            teacher.aux_lib_kind = hd_s.precision
        if hd_s.L is not hd_t.L or (not t_split and hd_t.precision != hd_s.precision):
            if not getattr(self, "_warned_tail_lib", False):
                self._warned_tail_lib = True
                import warnings
                warnings.warn("fused optimizer tail refused: the student's and the teacher's plans live in different library builds "
                              f"({hd_s.precision} / {hd_t.precision}); falling back to Adam + EMA + pack launches for this step")
            return False
        if len(group["params"]) != len(ps) or any(a is not b for a, b in zip(group["params"], ps)):
            return False
        if len(ema.source_params) != len(ps) or any(a is not b for a, b in zip(ema.source_params, ps)):
            return False
        got = self._gather(0, group)
        if got is None:
            return False
        with_grad, tab = got
        ent = self._dev_state(0, group, ps[0].device)
        tps = ema.target_params
        key = (ps[0].data_ptr(), tps[0].data_ptr(), with_grad[0].grad.data_ptr(), self.state[with_grad[0]]["exp_avg"].data_ptr(),
               hd_s.wpack.data_ptr(), hd_t.wpack.data_ptr(), len(with_grad))
        cache = getattr(self, "_tail", None)
        if cache is None or cache[0] != key:
            n = len(ps)
            arr = lambda vals: (C.c_void_p * n)(*vals)
            st = [self.state[p] if p.grad is not None else None for p in ps]
            cache = (key, arr([p.data_ptr() for p in ps]), arr([p.grad.data_ptr() if p.grad is not None else None for p in ps]),
                     arr([s_["exp_avg"].data_ptr() if s_ else None for s_ in st]), arr([s_["exp_avg_sq"].data_ptr() if s_ else None for s_ in st]),
                     arr([p.data_ptr() for p in tps]))
            rc = hd_s.L.udapose_net_bind_update(hd_s.h, hd_t.h, cache[1], cache[2], cache[3], cache[4], cache[5], ptr(hd_s.wpack), ptr(hd_t.wpack))
            if rc == -3:        # unsupported layout (unaligned tensors, channel counts that are not multiples of 64)
                self._tail = None
                return False
            check(rc, "net_bind_update")
            self._tail = cache
        _, pa_s, ga, ma, va, pa_t = cache
        group["step"] = group.get("step", 0) + 1
        b1, b2 = group["betas"]
        if not torch.cuda.is_current_stream_capturing():
            self.sync_hyper()
        # a pending sum of the two passes' gradient buffers is taken in the sweep itself (no separate axpy over 220 MB) - and by the loss
        # scaler's inf / nan check in front of it, which looks at g + g2 (round 4: the fp16 step used to force the sum first)
        delta = student.pending_grad_sum(take=True) if hasattr(student, "pending_grad_sum") else 0
        self._pre_sweep(tab, ent, delta)
        args = (hd_s.h, hd_t.h, None, pa_s, ga, ma, pa_t, ptr(hd_s.wpack), ptr(hd_t.wpack), float(group["lr"]),
                float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]), int(group["step"]),
                float(group.get("grad_scale", 1.0)), ptr(ent[0]), float(ema.alpha), float(1.0 - ema.alpha), 1, int(delta))
        check(hd_s.L.udapose_net_fused_update(*(args[:2] + (_hip.stream(),) + args[3:])), "net_fused_update")
        return self._tail_finish(student, teacher, hd_s, hd_t, ps, tps, t_split, ent)

    def _tail_finish(self, student, teacher, hd_s, hd_t, ps, tps, t_split, ent):
        self._post_sweep(ent)
        _bump_versions(ps)
        _bump_versions(tps)
        student.packs_refreshed(hd_s, True)
        if t_split:
            pa_t_ = teacher._pointers()[0]
            check(hd_t.L.udapose_net_pack_weights(hd_t.h, _hip.stream(), pa_t_, ptr(hd_t.wpack), 0), "net_pack_weights (teacher, f16x2)")
        teacher.packs_refreshed(hd_t, False)
        return True


class FusedSGD(_FusedBase):
    _state_names = ("momentum_buffer",)

    def __init__(self, params, lr, momentum=0.9, weight_decay=0.0, nesterov=False, grad_scale=1.0, dynamic_loss_scale=False,
                 init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000):
        scaler = dict(init_scale=init_scale, growth_factor=growth_factor, backoff_factor=backoff_factor,
                      growth_interval=growth_interval) if dynamic_loss_scale else None
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay, nesterov=nesterov, grad_scale=grad_scale, step=0),
                         scaler)

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        capturing = torch.cuda.is_current_stream_capturing()
        for gi, group in enumerate(self.param_groups):
            got = self._gather(gi, group)
            if got is None:
                continue
            ps, t = got
            ent = self._dev_state(gi, group, ps[0].device)
            group["step"] = group.get("step", 0) + 1
            if not capturing:
                self.sync_hyper()
            self._pre_sweep(t, ent)
            check(lib().udapose_sgd_multi(_hip.stream(), ptr(t.ptrs[0]), ptr(t.ptrs[1]), ptr(t.ptrs[2]), ptr(t.sizes), ptr(t.blk_t), ptr(t.blk_o),
                                          t.nblocks, float(group["lr"]), float(group["momentum"]), float(group["weight_decay"]),
                                          int(group["nesterov"]), int(group["step"] == 1), float(group.get("grad_scale", 1.0)), ptr(ent[0])),
                  "sgd_multi")
            self._post_sweep(ent)
            _bump_versions(ps)
        return loss
