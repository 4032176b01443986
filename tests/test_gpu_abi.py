"""The C-ABI contract of include/udapose.h: explicit dispatch policy instead of environment variables or
debug-setter globals, explicit preparation (bind / prepare) so that compute calls never allocate, distinct plans re-entrant
from different host threads on different streams."""
import threading

import pytest
import torch

pytestmark = pytest.mark.gpu



@pytest.fixture(autouse=True)
def _bf16_unless_stated(monkeypatch):
    """The tests of this module exercise the 16-bit executor: networks start in 'bf16' (BASELINE.json's benched precision) unless a
    test sets another precision.  (A new module's default is 'auto': autocast dtype / fp32-grade teacher, tests/test_gpu_dropin_loop.py.)"""
    from uda_poseestimation_amd.lib.models.pose_resnet import PoseResNet
    monkeypatch.setattr(PoseResNet, "default_precision", "bf16")

def _tiny(K=16, seed=0):
    import uda_poseestimation_amd.lib.models.pose_resnet as pr
    torch.manual_seed(seed)
    return pr._pose_resnet("t", K, pr.Bottleneck_default, [1, 1, 1, 1], False, False)


def test_two_threads_two_plans_two_streams():
    """Two host threads drive two networks (two executor plans) on two streams at the same time, forward + backward, several
    rounds; every thread gets exactly what it gets when it runs alone."""
    nets = [_tiny(seed=s).cuda() for s in (1, 2)]
    xs = [torch.randn(4, 3, 128, 128, generator=torch.Generator().manual_seed(10 + i)).cuda() for i in range(2)]
    Rs = [torch.randn(4, 16, 32, 32, generator=torch.Generator().manual_seed(20 + i)).cuda() for i in range(2)]

    def run(i, out, rounds):
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            for _ in range(rounds):
                nets[i].zero_grad(set_to_none=True)
                y = nets[i](xs[i])
                (y * Rs[i]).sum().backward()
            st.synchronize()
        out[i] = (y.detach().clone(), {n: p.grad.clone() for n, p in nets[i].named_parameters() if p.grad is not None})

    def reset():
        for i, s in enumerate((1, 2)):
            nets[i].load_state_dict(_tiny(seed=s).state_dict())     # same weights AND fresh BN running statistics

    serial = {}
    for i in range(2):
        run(i, serial, 3)
    reset()
    both = {}
    th = [threading.Thread(target=run, args=(i, both, 3)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for i in range(2):
        assert torch.equal(serial[i][0], both[i][0]), f"thread {i}: heat-maps differ from the serial run"
        for n in serial[i][1]:
            a, b = serial[i][1][n], both[i][1][n]
            # (split weight-gradient reductions add with fp32 atomics: order, hence the last bits, may differ)
            assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max()) + 1e-9, (i, n)
        for (n1, b1), (n2, b2) in zip(nets[i].named_buffers(), _tiny(seed=i + 1).named_buffers()):
            if "num_batches_tracked" in n1:
                assert int(b1) == 3


def test_compute_calls_never_build_tables():
    """A convolution geometry whose tap tables were not prepared cannot be run inside a stream capture (the call refuses
    instead of allocating); after udapose_conv_prepare the same capture works.  A network plan refuses pointers it was
    not bound to."""
    import ctypes as C
    from uda_poseestimation_amd import _hip, ops
    lib = _hip.lib()
    N, H, Ci, Co = 1, 8, 64, 64
    d = ops.conv_desc(N, H, H, Ci, Co, 5, 1, 2)            # 5x5: a geometry class nothing else in the suite uses
    x = torch.randn(N, H, H, Ci, device="cuda").bfloat16()
    w = torch.randn(Co, 25, Ci, device="cuda").bfloat16()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        with pytest.raises(RuntimeError, match="not prepared"):
            ops.conv2d_fwd(x, w, d)
    assert lib.udapose_conv_prepare(C.byref(d)) == 0
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        y = ops.conv2d_fwd(x, w, d)
    g2.replay()
    torch.cuda.synchronize()
    ref = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w.float().reshape(Co, 5, 5, Ci).permute(0, 3, 1, 2), padding=2)
    assert float((y.float().permute(0, 3, 1, 2) - ref).abs().max()) <= 2e-2 * float(ref.abs().max())
    # a plan only runs on what it was bound to
    net = _tiny().cuda()
    xin = torch.randn(2, 3, 64, 64, device="cuda")
    net(xin).sum().backward()                                # (binds parameters, buffers, pack and the gradient placement)
    hd = next(iter(net._handles.values()))
    other = (C.c_void_p * hd.n_params)(*[p.data_ptr() + 0 for p in reversed(list(net.parameters()))])
    rc = hd.L.udapose_net_pack_weights(hd.h, _hip.stream(), other, _hip.ptr(hd.wpack), 1)
    assert rc == -4                                          # UDAPOSE_ERR_NOT_PREPARED, nothing allocated behind the caller's back


def test_policy_is_explicit_and_per_plan():
    """The default policy is the production one; a plan's policy is what was set on IT (no process-wide switch)."""
    import ctypes as C
    from uda_poseestimation_amd import _hip
    p = _hip.policy()
    assert (p.igemm_tile, p.igemm_h3, p.wgrad_group, p.wgrad_stages, p.bn_bwd_fused, p.debug_sync) == (-1, 1, 1, 128, 1, 0)
    with pytest.raises(KeyError):
        _hip.policy(no_such_field=1)
    a, b = _tiny(seed=1).cuda(), _tiny(seed=1).cuda()
    b.policy = {"wgrad_group": 0, "bn_bwd_fused": 0, "igemm_h3": 0}
    x = torch.randn(2, 3, 64, 64, device="cuda")
    ya, yb = a(x), b(x)
    for net, expect in ((a, (1, 1, 1)), (b, (0, 0, 0))):
        hd = next(iter(net._handles.values()))
        got = _hip.Policy()
        assert hd.L.udapose_net_get_policy(hd.h, C.byref(got)) == 0
        assert (got.wgrad_group, got.bn_bwd_fused, got.igemm_h3) == expect
    assert float((ya - yb).abs().max()) <= 2e-2 * float(ya.abs().max())      # same function, other kernels
    ya.sum().backward(); yb.sum().backward()
    for (n, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        if pa.grad is not None:
            assert torch.isfinite(pb.grad).all(), n


def test_error_codes_of_the_round2_entry_points():
    """Argument validation at the C ABI: bad arguments return UDAPOSE_ERR_ARG (-1) before anything is enqueued - null pointers,
    row sizes that are not 16-byte multiples, an unknown backward phase, a phase split without grouped weight gradients."""
    import ctypes as C
    from uda_poseestimation_amd import _hip
    L = _hip.lib()
    s = torch.cuda.current_stream().cuda_stream
    conf = torch.zeros(4, 16, device="cuda")
    idx = torch.zeros(4, 16, dtype=torch.int32, device="cuda")
    u = torch.zeros(4, 4, device="cuda")
    boxes = torch.zeros(4, 6, dtype=torch.int32, device="cuda")
    ap = torch.zeros(4, dtype=torch.uint8, device="cuda")
    p = lambda t: t.data_ptr()
    assert L.udapose_occlusion_pick(s, None, p(idx), p(u), 4, 16, 64, 4.0, 256, 0.5, 0.9, 10, p(boxes), p(ap)) == -1
    assert L.udapose_occlusion_pick(s, p(conf), p(idx), p(u), 0, 16, 64, 4.0, 256, 0.5, 0.9, 10, p(boxes), p(ap)) == -1
    assert L.udapose_occlusion_pick(s, p(conf), p(idx), p(u), 4, 16, 64, 4.0, 256, 0.5, 0.9, 10, p(boxes), p(ap)) == 0
    torch.cuda.synchronize()
    assert int(ap.sum()) == 0 and int(boxes.abs().sum()) == 0            # (no confidence reaches 0.9: nothing selected, zero-area boxes)
    a, b, d = (torch.zeros(4, 30, device="cuda") for _ in range(3))
    assert L.udapose_select_rows(s, p(d), p(a), p(b), p(ap), 4, 30) == -1            # 30 floats: not a multiple of 16 bytes
    assert L.udapose_select_rows(s, p(d), p(a), None, p(ap), 4, 32) == -1
    x = torch.zeros(1, 4, 4, 64, dtype=torch.bfloat16, device="cuda")
    assert L.udapose_adain_alpha_dev(s, p(x), p(x), p(x), 1, 16, 16, 64, 1e-5, None, None, 0) == -1
    net = _tiny().cuda().train()
    xin = torch.randn(2, 3, 64, 64, device="cuda")
    y = net(xin)
    y.sum().backward()                                                   # (plans, tables and gradient placement now exist)
    hd = net._last_hd
    pa, ba, params = net._pointers()
    act = torch.empty(hd.act_bytes, dtype=torch.uint8, device="cuda")
    ws = torch.empty(hd.ws.numel(), dtype=torch.uint8, device="cuda")
    gp = net._grad_ptrs[1]
    args = (hd.h, s, None, pa, p(hd.wpack), p(act), p(ws), gp, C.c_float(0.0))
    assert hd.L.udapose_net_backward_phase(*args, 1, 3) == -1            # unknown phase
    assert hd.L.udapose_net_backward_phase(*args, 3, 0) == -1            # unknown part
    assert hd.L.udapose_net_backward_phase(None, s, None, pa, p(hd.wpack), p(act), p(ws), gp, C.c_float(0.0), 1, 1) == -1
