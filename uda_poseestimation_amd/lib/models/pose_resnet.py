"""Simple-Baseline pose network (API mirror of the reference's lib/models/pose_resnet.py:11-126) on the MI355X executor.

`PoseResNet.forward` = head(upsampling(backbone(x))) exactly as pose_resnet.py:80-84, but the whole chain (107 convs /
BNs, ReLUs, max-pool, 3 deconvs, head) is enqueued by ONE call into libudapose_hip.so (csrc/net.hip); backward is one
call as well.  Sub-modules are the usual torch.nn containers, so `state_dict()` keys, `.parameters()` order, `.cuda()`,
`.train()/.eval()`, optimizers, GradScaler and OldWeightEMA behave as with the reference.  There is no CPU path.
"""
import ctypes as C
import weakref

import torch
import torch.nn as nn

from ... import _hip
from ..._hip import check, lib, ptr
from .resnet import _resnet
from .resnet import Bottleneck as Bottleneck_default

__all__ = ['pose_resnet101', 'pose_resnet50']


class Upsampling(nn.Sequential):
    """3-layer deconvolution of Simple Baseline (pose_resnet.py:11-56); containers only, executed by the net executor."""

    def __init__(self, in_channel=2048, hidden_dims=(256, 256, 256), kernel_sizes=(4, 4, 4), bias=False):
        assert len(hidden_dims) == len(kernel_sizes), 'ERROR: len(hidden_dims) is different len(kernel_sizes)'
        layers = []
        for hidden_dim, kernel_size in zip(hidden_dims, kernel_sizes):
            if kernel_size == 4:
                padding, output_padding = 1, 0
            elif kernel_size == 3:
                padding, output_padding = 1, 1
            elif kernel_size == 2:
                padding, output_padding = 0, 0
            else:
                raise NotImplementedError("kernel_size is {}".format(kernel_size))
            layers.append(nn.ConvTranspose2d(in_channel, hidden_dim, kernel_size, stride=2, padding=padding,
                                             output_padding=output_padding, bias=bias))
            layers.append(nn.BatchNorm2d(hidden_dim))
            layers.append(nn.ReLU(inplace=True))
            in_channel = hidden_dim
        super().__init__(*layers)
        for m in self.modules():
            if isinstance(m, nn.ConvTranspose2d):
                nn.init.normal_(m.weight, std=0.001)
                if bias:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        self._cfg = (tuple(hidden_dims), tuple(kernel_sizes), bool(bias))


class _NetHandle:
    """One executor plan (fixed N,H,W) plus its device work areas."""

    def __init__(self, layers, K, N, H, W, device, precision='bf16', policy=None, deconv_bias=False, aux_lib='bf16', fwd_only=False):
        # the library build whose element type is this plan's storage / MFMA type.  'fp32' / 'f16x2' plans do not depend on the
        # element type (both builds carry them): they live in the build `aux_lib` names - the STUDENT's build when this network is
        # the teacher of a fused optimizer tail, which hands both plans to one library (a plan is only ever
        # dereferenced by the .so that created it)
        L = self.L = lib('fp16' if precision == 'fp16' else ('bf16' if precision == 'bf16' else aux_lib))
        h = C.c_void_p()
        arr = (C.c_int * 4)(*layers)
        # fwd_only (mode bit 9): a plan for no-grad forwards - the teacher, validate() - whose y / z tensors rotate through six scratch
        # buffers instead of a 2.8 GB arena (udapose.h); it refuses a backward
        self.fwd_only = bool(fwd_only)
        check(L.udapose_net_create(arr, K, N, H, W, {'fp32': 1, 'f16x2': 2}.get(precision, 0) | (0x100 if deconv_bias else 0) | (0x200 if fwd_only else 0),
                                   C.byref(h)), "net_create")
        self.h = h
        self.precision = precision
        if policy:
            pol = _hip.policy(**policy)
            check(L.udapose_net_set_policy(h, C.byref(pol)), "net_set_policy")
        self.bound = None            # pointer key the plan's device tables were built for (udapose_net_bind)
        self.bound_grads = set()     # gradient placements bound so far (udapose_net_bind_grads)
        self.n_params = L.udapose_net_num_params(h)
        self.n_buffers = L.udapose_net_num_buffers(h)
        self.numel = [L.udapose_net_param_numel(h, i) for i in range(self.n_params)]
        self.act_bytes = L.udapose_net_act_bytes(h)
        shp = (C.c_int * 4)()
        L.udapose_net_out_shape(h, shp)
        self.out_shape = tuple(shp)
        self.ws = torch.empty(L.udapose_net_ws_bytes(h), dtype=torch.uint8, device=device)
        self.wpack = torch.empty(L.udapose_net_wpack_bytes(h), dtype=torch.uint8, device=device)
        self.wpack_version = None
        # (the backward can be cut at layer3's first block, and its weight gradients are grouped launches)
        self.grouped = int((policy or {}).get("wgrad_group", 1)) != 0
        self.can_split = L.udapose_net_grad_split_param(h) >= 0 and self.grouped
        self.act_nograd = None
        self._fin = weakref.finalize(self, L.udapose_net_destroy, h)


class _PoseNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, net, need_grad, defer_bn, *params):
        # (grad mode is always off inside Function.forward: the caller decides whether backward state is kept)
        out, act, hd, ws = net._run_forward(x, save=need_grad, defer_bn=defer_bn)
        ctx.net, ctx.act, ctx.hd, ctx.ws = net, act, hd, ws
        ctx.nparams = len(params)
        return out

    @staticmethod
    def backward(ctx, dout):
        net = ctx.net
        if ctx.act is None:
            raise RuntimeError("PoseResNet backward without saved activations (forward ran without grad, or backward ran twice)")
        net._run_backward(dout, ctx.act, ctx.hd, ctx.ws)
        ctx.act = ctx.ws = None
        # parameter gradients are accumulated straight into p.grad (views of the module's flat gradient buffer)
        return (None, None, None, None) + (None,) * ctx.nparams


class PoseResNet(nn.Module):
    """Simple Baseline for key-point detection (pose_resnet.py:59-91) on the MI355X executor."""
    default_precision = 'auto'      # what a new module's `precision` starts as (see __init__)
    _warned_bf16_fallback = False
    fwd_only_plans = True           # no-grad forwards (teacher, validate()) run forward-only plans: y / z in six rotating scratch buffers

    def __init__(self, backbone, upsampling, feature_dim, num_keypoints, finetune=False):
        super().__init__()
        self.backbone = backbone
        self.upsampling = upsampling
        self.head = nn.Conv2d(in_channels=feature_dim, out_channels=num_keypoints, kernel_size=1, stride=1, padding=0)
        self.finetune = finetune
        for m in self.head.modules():
            nn.init.normal_(m.weight, std=0.001)
            nn.init.constant_(m.bias, 0)
        cfg = getattr(upsampling, "_cfg", None)
        if cfg is None or cfg[:2] != ((256, 256, 256), (4, 4, 4)) or feature_dim != 256:
            raise NotImplementedError("the MI355X executor implements the reference configuration: 3 x deconv(256, k=4)")
        self._deconv_bias = bool(cfg[2])          # deconv_with_bias (pose_resnet.py:15,41,96)
        self.num_keypoints = num_keypoints
        self.bn_momentum = 0.1
        # 'auto' (default): what the reference's scripts get from torch - inside `torch.cuda.amp.autocast()` the autocast dtype
        #         (fp16 by default: the student of train_human.py:280,414; bf16 if the context names it); outside autocast, a
        #         forward that keeps no backward state (the teacher under no_grad, validate(): train_human.py:346-358,461-500 run in
        #         fp32) takes the fp32-grade 'f16x2' mode; a differentiable forward outside autocast takes 'bf16' (there is no fp32
        #         backward: the reference's scripts always train under autocast).
        #         (memory: every (N, H, W, precision) a module is called with owns an executor plan - weight packs, a workspace and, for no-grad
        #         forwards, one activation arena: 88 MB per image at 256x256 in the 16-bit modes, twice that in f16x2 - so an 'auto' student
        #         that is also evaluated under no_grad holds a second, f16x2 plan of its batch size: 5.6 GB at N = 32; `del model._handles[key]`
        #         or `model._handles.clear()` releases plans that are no longer needed.)
        # 'bf16': bf16 storage + MFMA, fp32 accumulation (training and inference; BASELINE.json's benched precision).
        # 'fp16': fp16 storage + MFMA (v_mfma_f32_16x16x32_f16), fp32 accumulation - the reference's autocast dtype
        #         (train_human.py:280,414); gradients need loss scaling (GradScaler, or optim.FusedAdam(dynamic_loss_scale=True)).
        # 'f16x2': the FAST fp32-grade mode, forward only: activations and weights as fp16 pairs (h, l), three fp16 MFMAs per K step,
        #         fp32 accumulation / BatchNorm statistics - heat-maps within ~4e-5 of the fp32 CPU oracle, 1.7x the exact mode's speed.
        # 'fp32': exact fp32 MFMA (v_mfma_f32_16x16x4_f32), forward only; 3x slower than bf16.
        self.precision = type(self).default_precision
        self.aux_lib_kind = 'bf16'    # build that holds this module's 'fp32' / 'f16x2' plans (engine: the student's build, see _NetHandle)
        self._handles = {}
        self._ptr_cache = None
        self._flat_grad = None
        self._flat_grad2 = None       # second per-pass gradient buffer (backward passes running on different streams)
        self._grad_state = None       # (stream of the first backward of this step, pending second-buffer sum?)
        self._deferred_bn = []        # forwards whose BN running-statistics update is still to be applied, in call order
        # Weights epoch: bumped by whoever changes parameter VALUES without torch seeing it (hipGraph replays of the fused
        # Adam / EMA kernels): the bf16 weight packs of EVERY executor plan are stale once it moves.  In-place torch ops and
        # the eager fused optimizers are caught by the parameters' version counters as well.
        self._wepoch = 0
        self._capture_token = None    # set by GraphedTrainStep around a capture: pack once per capture, unconditionally
        # explicit dispatch-policy overrides for this network's executor plans (fields of udapose_policy, include/udapose.h);
        # empty = the production policy.  Plans read it when they are created: clear self._handles after changing it.
        self.policy = {}
        # split_backward = True (data parallel): cut every backward after the first block of layer3 (udapose_net_backward_part).  backward() then runs
        # part 1 only - after it the gradients of layer3 / layer4 / upsampling / head (a contiguous suffix of the flat buffer,
        # 94 % of it) are final - and finish_backward() runs part 2; the caller all-reduces the suffix in between, under part 2.
        self.split_backward = False
        self._pending_lower = []
        # merge_wgrad (set by the engine around a step's backward): backward() runs the gradient chain only and finish_wgrad() then
        # launches the grouped weight gradients - of BOTH passes in one grid when two passes of the same plan are pending
        # (udapose_net_wgrad_pair): the passes end together and their weight-gradient launches are exposed at the step's end
        self.merge_wgrad = False
        self._pending_wg = []
        self._to_channels_last()

    # ------------------------------------------------------------------ layout / pointer bookkeeping
    def _to_channels_last(self):
        for p in self.parameters():
            if p.dim() == 4 and not p.data.is_contiguous(memory_format=torch.channels_last):
                p.data = p.data.contiguous(memory_format=torch.channels_last)
            elif p.dim() == 4 and p.data.stride(1) != 1:
                p.data = p.data.contiguous(memory_format=torch.channels_last)

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._to_channels_last()
        self._ptr_cache = None
        self._flat_grad = None
        self._flat_grad2 = None
        self._grad_state = None
        self._deferred_bn = []
        self._handles = {}
        self._pending_lower = []
        self._pending_wg = []
        self._split_off = None
        return r

    def load_state_dict(self, state_dict, strict=True, **kw):
        # accept checkpoints saved from DataParallel wrappers ('module.' prefix, train_human.py:229-230)
        if any(k.startswith("module.") for k in state_dict):
            state_dict = {(k[7:] if k.startswith("module.") else k): v for k, v in state_dict.items()}
        r = super().load_state_dict(state_dict, strict=strict, **kw)
        if kw.get("assign"):
            self._ptr_cache = None          # (Parameter objects were replaced: pointer / version caches are void)
            self._to_channels_last()
            self._handles = {}
        return r

    def _pointers(self):
        params = list(self.parameters())
        bufs = list(self.buffers())
        key = (params[0].data_ptr(), params[-1].data_ptr(), bufs[0].data_ptr(), len(params))
        if self._ptr_cache is None or self._ptr_cache[0] != key:
            for p in params:
                if p.dtype != torch.float32:
                    raise RuntimeError("PoseResNet master parameters must be fp32 (bf16 compute copies are made by the executor)")
                if p.dim() == 4 and p.stride(1) != 1 and p.shape[1] != 1:
                    self._to_channels_last()
                    return self._pointers()
            pa = (C.c_void_p * len(params))(*[p.data_ptr() for p in params])
            ba = (C.c_void_p * len(bufs))(*[b.data_ptr() for b in bufs])
            self._ptr_cache = (key, pa, ba, params)
        return self._ptr_cache[1], self._ptr_cache[2], self._ptr_cache[3]

    def _grad_views(self, params):
        """p.grad tensors are views (same strides as p) of one flat fp32 buffer: one all-reduce / one Adam sweep."""
        if self._flat_grad is None or self._flat_grad.device != params[0].device:
            total = sum(p.numel() for p in params)
            self._flat_grad = torch.zeros(total, dtype=torch.float32, device=params[0].device)
            views, off = [], 0
            for p in params:
                views.append(self._flat_grad[off:off + p.numel()].as_strided(p.shape, p.stride()))
                off += p.numel()
            self._grad_view_list = views
        return self._grad_view_list

    def resolved_precision(self, differentiable):
        """The precision a forward started now would run in ('auto' resolved, see __init__)."""
        prec = self.precision
        if prec not in ('auto', 'bf16', 'fp16', 'fp32', 'f16x2'):
            raise ValueError("precision must be 'auto', 'bf16', 'fp16', 'f16x2' or 'fp32'")
        if prec != 'auto':
            return prec
        if torch.is_autocast_enabled():
            return 'bf16' if torch.get_autocast_dtype('cuda') == torch.bfloat16 else 'fp16'
        if differentiable and not PoseResNet._warned_bf16_fallback:
            PoseResNet._warned_bf16_fallback = True
            import warnings
            warnings.warn("PoseResNet(precision='auto'): a differentiable forward OUTSIDE torch.autocast runs in bf16 here (the reference would run "
                          "fp32: there is no fp32 backward on this path; its scripts always train under autocast, train_human.py:280,414). Set "
                          ".precision explicitly ('bf16' / 'fp16') to silence this.", stacklevel=3)
        return 'bf16' if differentiable else 'f16x2'

    def _handle(self, x, differentiable=None):
        N, Cc, H, W = x.shape
        if Cc != 3:
            raise ValueError("PoseResNet expects [N,3,H,W] input")
        if differentiable is None:
            differentiable = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        prec = self.resolved_precision(differentiable)
        aux = self.aux_lib_kind if prec in ('fp32', 'f16x2') else None
        fo = bool(PoseResNet.fwd_only_plans) and not differentiable
        key = (N, H, W, x.device.index, prec, aux, fo)
        hd = self._handles.get(key)
        if hd is None:
            hd = _NetHandle(self.backbone.layers_cfg, self.num_keypoints, N, H, W, x.device, prec, dict(self.policy), self._deconv_bias,
                            aux_lib=(aux or 'bf16'), fwd_only=fo)
            params = list(self.parameters())
            assert hd.n_params == len(params) and hd.n_buffers == len(list(self.buffers())), "executor/module parameter mismatch"
            for i, p in enumerate(params):
                assert p.numel() == hd.numel[i], f"parameter {i} size mismatch: {p.numel()} vs {hd.numel[i]}"
            self._handles[key] = hd
        return hd

    # ------------------------------------------------------------------ executor calls
    def prepare(self, x):
        """Create the executor plan for x's shape and refresh the bf16 weight packs on the current stream (so that forwards
        launched afterwards on OTHER streams only need to wait for this point)."""
        hd = self._handle(x)
        pa, ba, params = self._pointers()
        self._pack(hd, pa, params, need_bwd=torch.is_grad_enabled() and any(p.requires_grad for p in params))
        return hd

    def weights_changed(self):
        """Tell the executor that parameter values changed behind torch's back (raw-pointer kernels replayed from a
        hipGraph): every plan re-packs its bf16 weights before its next forward."""
        self._wepoch += 1

    def version_key(self):
        """What the weight packs of a plan are valid for: the weights epoch and the parameters' version counters."""
        # (the cached parameter list of _pointers(): walking the module tree costs ~0.4 ms per call on PoseResNet-101, and a
        # captured step asks four times per replay)
        plist = self._ptr_cache[3] if self._ptr_cache is not None else list(self.parameters())
        return (self._wepoch, sum(p._version for p in plist))

    def __setattr__(self, name, value):
        # a Parameter object replaced behind the cached parameter list (module.x = nn.Parameter(...), load_state_dict(assign=True))
        # must not leave version_key() watching the old objects
        if isinstance(value, (nn.Parameter, nn.Module)) and "_ptr_cache" in self.__dict__:
            self.__dict__["_ptr_cache"] = None
        super().__setattr__(name, value)

    def packs_refreshed(self, hd, with_bwd):
        """A kernel outside _pack (the fused optimizer tail) has just rewritten hd's packs from the current parameters."""
        hd.wpack_version = (self.version_key(), bool(with_bwd))
        hd.maintained = True

    def _bind(self, hd, pa, ba):
        """Build the plan's device tables for the current parameter / buffer pointers (allocates: never inside a capture; the
        compute calls themselves never allocate and fail with 'not prepared' otherwise)."""
        key = self._ptr_cache[0]
        if hd.bound != key:
            check(hd.L.udapose_net_bind(hd.h, pa, ba, ptr(hd.wpack)), "net_bind")
            hd.bound, hd.bound_grads = key, set()

    def _pack(self, hd, pa, params, need_bwd):
        self._bind(hd, pa, self._ptr_cache[2])
        version = (self._wepoch, sum(p._version for p in params))
        if torch.cuda.is_current_stream_capturing():
            # inside a capture the pack kernels must be IN the graph (a replay runs on the weights of that moment, whatever
            # the host-side cache says): once per capture when the capturer handed out a token, on every call otherwise
            tok = self._capture_token
            cap_tok, cap_bwd = getattr(hd, "_cap", (None, False))
            same = tok is not None and cap_tok is tok
            if same and (cap_bwd or not need_bwd):
                return
            if tok is not None and getattr(hd, "maintained", False) and hd.wpack_version in ((version, need_bwd), (version, True)):
                # the captured step ends with the fused optimizer tail, which rewrites this plan's packs: they are valid now and
                # after every replay (GraphedTrainStep.step re-packs eagerly if anything else touched the weights in between)
                return
            check(hd.L.udapose_net_pack_weights(hd.h, _hip.stream(), pa, ptr(hd.wpack), int(need_bwd)), "net_pack_weights")
            hd._cap = (tok, bool(need_bwd) or (same and cap_bwd))
            hd.wpack_version = None       # what a replay leaves in the pack is unknown to the host-side cache
            return
        if hd.wpack_version != (version, need_bwd) and hd.wpack_version != (version, True):
            check(hd.L.udapose_net_pack_weights(hd.h, _hip.stream(), pa, ptr(hd.wpack), int(need_bwd)), "net_pack_weights")
            hd.wpack_version = (version, need_bwd)

    def _run_forward(self, x, save, defer_bn=False):
        _hip.require_cuda(x)
        if x.dtype != torch.float32:
            x = x.float()
        x = x.contiguous()
        if save and not self.training:
            # eval-mode BatchNorm is folded into the convolutions' epilogues (policy eval_fold): such a forward writes neither the pre-BN
            # tensors nor the batch statistics a backward would read, and there is no eval-mode BN backward on this path.
            # An eval-mode forward outside torch.no_grad() (a validation loop that forgot it) therefore keeps NO backward state; a later
            # .backward() raises in _PoseNetFn.backward ("backward without saved activations") instead of reading unwritten memory
            save = False
        hd = self._handle(x, differentiable=save)
        self._last_hd = hd            # the plan of the most recent forward (the fused optimizer tail keeps ITS packs fresh)
        if save and hd.precision in ('fp32', 'f16x2'):
            raise RuntimeError(f"precision={hd.precision!r} is forward-only (run it under torch.no_grad(), as the reference does for the teacher)")
        pa, ba, params = self._pointers()
        s = _hip.stream()
        self._pack(hd, pa, params, need_bwd=save)
        if save:
            # every differentiable forward owns its activation arena AND its scratch workspace, so that two forward /
            # backward passes of the same module can run concurrently on different streams
            act = torch.empty(hd.act_bytes, dtype=torch.uint8, device=x.device)
            ws = torch.empty(hd.ws.numel(), dtype=torch.uint8, device=x.device)
        else:
            if hd.act_nograd is None:
                hd.act_nograd = torch.empty(hd.act_bytes, dtype=torch.uint8, device=x.device)
            act, ws = hd.act_nograd, hd.ws
        out = torch.empty(hd.out_shape, dtype=torch.float32, device=x.device)
        defer = bool(defer_bn) and self.training
        check(hd.L.udapose_net_forward(hd.h, s, ptr(x), pa, ba, ptr(hd.wpack), ptr(act), ptr(ws), ptr(out),
                                        int(self.training) | (2 if defer else 0), float(self.bn_momentum)), "net_forward")
        if defer:
            self._deferred_bn.append((hd, act))
        return out, (act if save else None), hd, (ws if save else None)

    def apply_deferred_bn(self):
        """Apply, in call order and on the current stream, the BN running-statistics updates of forwards that ran with
        `forward_deferred_bn` (the caller has already made this stream wait for those forwards)."""
        pa, ba, params = self._pointers()
        for hd, act in self._deferred_bn:
            check(hd.L.udapose_net_apply_running(hd.h, _hip.stream(), ptr(act), ba, float(self.bn_momentum)), "net_apply_running")
        self._deferred_bn = []

    def finish_grads(self, part=0, defer=False):
        """Sum the second per-pass gradient buffer into p.grad's buffer (only needed when two backward passes ran on
        different streams; the caller has already made the current stream wait for both).  part 1 / 2: only the suffix /
        prefix of the flat buffer that backward part 1 / part 2 produced (split_backward).
        defer (whole buffer only): leave the sum to a consumer that reads both buffers itself (the fused optimizer tail takes
        pending_grad_sum()); until then p.grad holds the first pass's share only - a later finish_grads() completes it."""
        st = self._grad_state
        if defer and part == 0:
            return
        if st is not None and st[1]:
            n, off = self._flat_grad.numel(), (self.grad_split_offset() if part else 0)
            lo, cnt = (0, n) if part == 0 else ((off, n - off) if part == 1 else (0, off))
            check(lib().udapose_axpy_f32(_hip.stream(), self._flat_grad.data_ptr() + 4 * lo, self._flat_grad2.data_ptr() + 4 * lo, cnt), "axpy")
        if part != 1:
            self._grad_state = None

    def pending_grad_sum(self, take=False):
        """Byte distance from p.grad's buffer to the second per-pass buffer whose sum is still pending (0: nothing pending)."""
        st = self._grad_state
        if st is None or not st[1]:
            return 0
        delta = self._flat_grad2.data_ptr() - self._flat_grad.data_ptr()
        if take:
            self._grad_state = None
        return delta

    def grad_split_offset(self):
        """Element offset into the flat gradient buffer where backward part 1's gradients start (layer3's first parameter)."""
        off = getattr(self, "_split_off", None)
        if off is None:
            hd = next(iter(self._handles.values()))
            idx = hd.L.udapose_net_grad_split_param(hd.h)
            params = list(self.parameters())
            if idx < 0 or idx >= len(params):
                raise RuntimeError("this network has no layer3 boundary to split the backward at")
            off = sum(p.numel() for p in params[:idx])
            if off % 4:
                raise RuntimeError("gradient split offset is not 16-byte aligned")
            self._split_off = off
        return off

    def finish_backward(self):
        """Run part 2 (layer2, layer1, stem and their weight gradients) of every backward that ran with split_backward, each
        on the stream its part 1 ran on."""
        pending, self._pending_lower = self._pending_lower, []
        pa, ba, params = self._pointers()
        for hd, act, ws, gptrs, beta, stream in pending:
            with torch.cuda.stream(stream):
                check(hd.L.udapose_net_backward_part(hd.h, stream.cuda_stream, None, pa, ptr(hd.wpack), ptr(act), ptr(ws), gptrs, beta, 2), "net_backward part 2")
                act.record_stream(stream)
                ws.record_stream(stream)

    def _run_backward(self, dout, act, hd, ws):
        pa, ba, params = self._pointers()
        views = self._grad_views(params)
        cur = torch.cuda.current_stream()
        second_buffer = False
        if params[0].grad is None:
            beta = 0.0
            self._grad_state = [cur, False]
        elif params[0].grad.data_ptr() == views[0].data_ptr():
            beta = 1.0
            st = self._grad_state
            if st is not None and st[0] != cur:
                # the first backward of this step is (possibly still) running on another stream: write this pass into
                # its own buffer and let finish_grads() add the two
                second_buffer, beta = True, (1.0 if st[1] else 0.0)
                st[1] = True
        else:   # foreign gradient tensors: adopt their values, then accumulate
            for p, v in zip(params, views):
                if p.grad is not None:
                    v.copy_(p.grad)
                else:
                    v.zero_()
            beta = 1.0
        ga = getattr(self, "_grad_ptrs", None)
        if ga is None or ga[0] != views[0].data_ptr():
            ga = (views[0].data_ptr(), (C.c_void_p * len(views))(*[v.data_ptr() for v in views]))
            self._grad_ptrs = ga
        gptrs = ga[1]
        if second_buffer:
            if self._flat_grad2 is None or self._flat_grad2.device != self._flat_grad.device:
                self._flat_grad2 = torch.zeros_like(self._flat_grad)
                off, arr = 0, []
                for p in params:
                    arr.append(self._flat_grad2.data_ptr() + 4 * off)
                    off += p.numel()
                self._grad_ptrs2 = (C.c_void_p * len(arr))(*arr)
            gptrs = self._grad_ptrs2
        gkey = gptrs[0]      # (placement key: the tables hold offsets relative to the first gradient tensor)
        if gkey not in hd.bound_grads:
            check(hd.L.udapose_net_bind_grads(hd.h, gptrs), "net_bind_grads")
            hd.bound_grads.add(gkey)
        dout = dout.contiguous().float()
        if self.split_backward:
            check(hd.L.udapose_net_backward_part(hd.h, _hip.stream(), ptr(dout), pa, ptr(hd.wpack), ptr(act), ptr(ws), gptrs, beta, 1), "net_backward part 1")
            self._pending_lower.append((hd, act, ws, gptrs, beta, cur))       # (keeps the arenas alive until part 2 has run)
        elif self.merge_wgrad and hd.grouped:
            check(hd.L.udapose_net_backward_phase(hd.h, _hip.stream(), ptr(dout), pa, ptr(hd.wpack), ptr(act), ptr(ws), gptrs, beta, 0, 1),
                  "net_backward gradient chain")
            self._pending_wg.append((hd, act, ws, gptrs, beta, cur))      # (keeps the arenas alive until the weight gradients have run)
        else:
            check(hd.L.udapose_net_backward(hd.h, _hip.stream(), ptr(dout), pa, ptr(hd.wpack), ptr(act), ptr(ws), gptrs, beta), "net_backward")
        # backbone.fc is not part of forward (resnet.py:21-40): like autograd in the reference, it gets NO gradient (None, not
        # zeros: SGD's weight decay and Adam must skip it exactly as torch.optim skips parameters without .grad)
        nograd = self._no_grad_ids()
        for p, v in zip(params, views):
            if p.requires_grad and id(p) not in nograd:
                p.grad = v

    def finish_wgrad(self):
        """Launch the grouped weight gradients of the backward passes that ran with merge_wgrad, on the current stream (the caller
        has made it wait for the streams those passes ran on): two pending passes of one plan go out as ONE launch per tile class."""
        pend, self._pending_wg = self._pending_wg, []
        if not pend:
            return
        s = _hip.stream()
        cur = torch.cuda.current_stream()
        pa, ba, params = self._pointers()
        if len(pend) == 2 and pend[0][0] is pend[1][0]:
            (hd, actA, wsA, gA, bA, _), (_, actB, wsB, gB, bB, _) = pend
            check(hd.L.udapose_net_wgrad_pair(hd.h, s, ptr(actA), ptr(wsA), gA, bA, ptr(actB), ptr(wsB), gB, bB, 0), "net_wgrad_pair")
        else:
            for hd, act, ws, gptrs, beta, _ in pend:
                check(hd.L.udapose_net_backward_phase(hd.h, s, None, pa, ptr(hd.wpack), ptr(act), ptr(ws), gptrs, beta, 0, 2), "net_backward weight gradients")
        for q in pend:
            q[1].record_stream(cur)
            q[2].record_stream(cur)

    def _no_grad_ids(self):
        fc = getattr(self.backbone, "fc", None)
        return {id(p) for p in fc.parameters()} if fc is not None else set()

    _warned_eval_grad = False

    def forward(self, x):
        params = list(self.parameters())
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        if need_grad and not self.training:
            # eval-mode BatchNorm is folded into the convolutions' epilogues (policy eval_fold) and this path has no eval-mode BN backward: such a
            # forward keeps no backward state.  Say so HERE - the output carries no grad_fn, so a later .backward() fails at once with torch's own
            # "does not require grad" instead of deep inside the executor (the reference's frozen-BN fine-tuning is not on this path)
            if not PoseResNet._warned_eval_grad:
                PoseResNet._warned_eval_grad = True
                import warnings
                warnings.warn("PoseResNet: an eval-mode forward with grad enabled is not differentiable on this path (eval-mode BatchNorm is folded into "
                              "the convolutions; there is no eval-mode BatchNorm backward): the output has no grad_fn.  Wrap evaluation in torch.no_grad(), "
                              "or call .train() for a differentiable forward.", stacklevel=2)
            with torch.no_grad():
                return _PoseNetFn.apply(x, self, False, False, *params)
        return _PoseNetFn.apply(x, self, need_grad, False, *params)

    def forward_deferred_bn(self, x):
        """Same as forward(), but the BN running statistics are not touched by this call: they are updated later, in call
        order, by apply_deferred_bn().  For forwards of one module that run concurrently on different streams (the batch
        statistics used for normalisation are per call either way)."""
        params = list(self.parameters())
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        return _PoseNetFn.apply(x, self, need_grad, True, *params)

    def get_parameters(self, lr=1.):
        return [
            {'params': self.backbone.parameters(), 'lr': 0.1 * lr if self.finetune else lr},
            {'params': self.upsampling.parameters(), 'lr': lr},
            {'params': self.head.parameters(), 'lr': lr},
        ]


def _pose_resnet(arch, num_keypoints, block, layers, pretrained_backbone, deconv_with_bias, finetune=False, progress=True, **kwargs):
    backbone = _resnet(arch, block, layers, pretrained_backbone, progress, **kwargs)
    upsampling = Upsampling(backbone.out_features, bias=deconv_with_bias)
    model = PoseResNet(backbone, upsampling, 256, num_keypoints, finetune)
    return model


def pose_resnet101(num_keypoints, pretrained_backbone=True, deconv_with_bias=False, finetune=False, progress=True, **kwargs):
    """Simple Baseline with a ResNet-101 backbone (pose_resnet.py:102-112)."""
    return _pose_resnet('resnet101', num_keypoints, Bottleneck_default, [3, 4, 23, 3], pretrained_backbone, deconv_with_bias, finetune,
                        progress, **kwargs)


def pose_resnet50(num_keypoints, pretrained_backbone=True, deconv_with_bias=False, finetune=False, progress=True, **kwargs):
    """Simple Baseline with a ResNet-50 backbone (pose_resnet.py:116-126)."""
    return _pose_resnet('resnet50', num_keypoints, Bottleneck_default, [3, 4, 6, 3], pretrained_backbone, deconv_with_bias, finetune,
                        progress, **kwargs)
