"""Dev experiment: the tile classes of the pair launch of the grouped weight gradients on ONE stream back to back, or on two streams
concurrently (one launch's tail filled by the other's work-groups).  Uses policy exp0 (class skip mask) - timing only."""
import sys, ctypes as C
sys.path.insert(0, ".")
import torch
import uda_poseestimation_amd.lib.models as models
from uda_poseestimation_amd import _hip
from uda_poseestimation_amd._hip import check, ptr
N = 32
net = models.pose_resnet101(num_keypoints=16, pretrained_backbone=False).cuda().train()
net.precision = "bf16"
x = torch.randn(N, 3, 256, 256, device="cuda")
hd = net.prepare(x)
pa, ba, params = net._pointers()
s = torch.cuda.current_stream().cuda_stream
out = torch.empty(hd.out_shape, dtype=torch.float32, device="cuda")
dout = torch.randn(hd.out_shape, device="cuda") * 1e-3
acts, wss, gps = [], [], []
views = net._grad_views(params)
flat2 = torch.zeros_like(net._flat_grad)
for k in range(2):
    act = torch.empty(hd.act_bytes, dtype=torch.uint8, device="cuda"); ws = torch.empty(hd.ws.numel(), dtype=torch.uint8, device="cuda")
    net._pack(hd, pa, params, need_bwd=True)
    check(hd.L.udapose_net_forward(hd.h, s, ptr(x), pa, ba, ptr(hd.wpack), ptr(act), ptr(ws), ptr(out), 3, 0.1), "fwd")
    if k == 0:
        gp = (C.c_void_p * len(views))(*[v.data_ptr() for v in views])
    else:
        off, arr = 0, []
        for p_ in params:
            arr.append(flat2.data_ptr() + 4 * off); off += p_.numel()
        gp = (C.c_void_p * len(arr))(*arr)
    check(hd.L.udapose_net_bind_grads(hd.h, gp), "bind")
    check(hd.L.udapose_net_backward(hd.h, s, ptr(dout), pa, ptr(hd.wpack), ptr(act), ptr(ws), gp, C.c_float(0.0)), "bwd")
    acts.append(act); wss.append(ws); gps.append(gp)
torch.cuda.synchronize()
def setpol(mask):
    pol = _hip.policy(exp0=mask)
    check(hd.L.udapose_net_set_policy(hd.h, C.byref(pol)), "set_policy")
def pair(stream):
    check(hd.L.udapose_net_wgrad_pair(hd.h, stream.cuda_stream, ptr(acts[0]), ptr(wss[0]), gps[0], C.c_float(0.0), ptr(acts[1]), ptr(wss[1]), gps[1], C.c_float(0.0), 0), "pair")
main, side = torch.cuda.current_stream(), torch.cuda.Stream()
def run_serial():
    setpol(0); pair(main)
def run_only(mask):
    setpol(mask); pair(main)
def run_two():
    side.wait_stream(main)
    setpol(2); pair(main)          # class 0 on main
    setpol(1); pair(side)          # class 1 on the side stream
    main.wait_stream(side)
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for rep in range(2):
    print(f"serial {timeit(run_serial):.1f} us | class 0 only {timeit(lambda: run_only(2)):.1f} | class 1 only {timeit(lambda: run_only(1)):.1f} | two streams {timeit(run_two):.1f}", flush=True)
