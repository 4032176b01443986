"""Dev tool: time the grouped weight-gradient launches of PoseResNet-101 (N=32, 256x256) alone: one forward + backward to fill the
arenas, then udapose_net_backward_phase(part 0, phase 2) = the two grouped launches, repeated."""
import sys, time, ctypes as C
sys.path.insert(0, ".")
import torch
import uda_poseestimation_amd.lib.models as models
from uda_poseestimation_amd._hip import check, ptr
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
POL = dict(kv.split("=") for kv in sys.argv[2:])
net = models.pose_resnet101(num_keypoints=16, pretrained_backbone=False).cuda().train()
net.policy.update({k: int(v) for k, v in POL.items()})
x = torch.randn(N, 3, 256, 256, device="cuda")
hd = net.prepare(x)
pa, ba, params = net._pointers()
act = torch.empty(hd.act_bytes, dtype=torch.uint8, device="cuda")
ws = torch.empty(hd.ws.numel(), dtype=torch.uint8, device="cuda")
out = torch.empty(hd.out_shape, dtype=torch.float32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
net._pack(hd, pa, params, need_bwd=True)
check(hd.L.udapose_net_forward(hd.h, s, ptr(x), pa, ba, ptr(hd.wpack), ptr(act), ptr(ws), ptr(out), 1, 0.1), "fwd")
views = net._grad_views(params)
gp = (C.c_void_p * len(views))(*[v.data_ptr() for v in views])
check(hd.L.udapose_net_bind_grads(hd.h, gp), "bind_grads")
dout = torch.randn(hd.out_shape, device="cuda") * 1e-3
check(hd.L.udapose_net_backward(hd.h, s, ptr(dout), pa, ptr(hd.wpack), ptr(act), ptr(ws), gp, C.c_float(0.0)), "bwd")
torch.cuda.synchronize()
# a second pass (own arenas, own gradient buffer) for the pair launch
act2, ws2 = torch.empty_like(act), torch.empty_like(ws)
check(hd.L.udapose_net_forward(hd.h, s, ptr(x), pa, ba, ptr(hd.wpack), ptr(act2), ptr(ws2), ptr(out), 3, 0.1), "fwd2")
flat2 = torch.zeros_like(net._flat_grad)
off, arr = 0, []
for p_ in params:
    arr.append(flat2.data_ptr() + 4 * off); off += p_.numel()
gp2 = (C.c_void_p * len(arr))(*arr)
check(hd.L.udapose_net_bind_grads(hd.h, gp2), "bind_grads2")
check(hd.L.udapose_net_backward(hd.h, s, ptr(dout), pa, ptr(hd.wpack), ptr(act2), ptr(ws2), gp2, C.c_float(0.0)), "bwd2")
torch.cuda.synchronize()
PAIR = True
def run():
    if PAIR:
        check(hd.L.udapose_net_wgrad_pair(hd.h, s, ptr(act), ptr(ws), gp, C.c_float(0.0), ptr(act2), ptr(ws2), gp2, C.c_float(0.0), 0), "wgrad pair")
    else:
        check(hd.L.udapose_net_backward_phase(hd.h, s, None, pa, ptr(hd.wpack), ptr(act), ptr(ws), gp, C.c_float(0.0), 0, 2), "wgrad group")
for _ in range(5): run()
torch.cuda.synchronize()
for rep in range(3):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): run()
    b.record(); torch.cuda.synchronize()
    print(f"grouped weight gradients of TWO passes in one pair launch, N={N} each: {a.elapsed_time(b) / 20 * 1e3:.1f} us", flush=True)
