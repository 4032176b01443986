"""Dev tool (round 5): sizing of ONE student chain over both domains' images (N = 64; BatchNorm statistics would have to stay per domain) against
the step's two concurrent N = 32 student chains.  Timing only: the N = 64 chain here normalises over all 64 images (same launches as per-domain
statistics).  Every phase is its own LINEAR hipGraph replayed alone on the chip (no branch -> hardware-queue lottery in the figure):
forward (arena kept), gradient chain (PoseResNet._run_backward with merge_wgrad), grouped weight gradients of that pass.
The step's own phase times for comparison (profiles/r5_ab_runs.txt 6): three N = 32 forwards together 6.7 ms (3.37 alone each), two gradient
chains together 6.4 ms (4.21 alone each), pair weight-gradient launch 2.45 ms.
usage: python tools/exp_batched_student.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import uda_poseestimation_amd.lib.models as models
from uda_poseestimation_amd.lib.models.pose_resnet import PoseResNet

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
PoseResNet.default_precision = "bf16"
torch.manual_seed(0)


def phases(N):
    net = models.pose_resnet101(16, pretrained_backbone=False).to(dev).train()
    x = torch.randn(N, 3, 256, 256, device=dev)
    cs = torch.cuda.Stream(device=dev)
    def f():
        return net._run_forward(x, save=True)
    def b(st):
        out, act, hd, ws = st
        for p in net.parameters():
            p.grad = None
        net.split_backward, net.merge_wgrad = False, True
        net._run_backward(torch.full_like(out, 1e-3), act, hd, ws)
        net.merge_wgrad = False
    def w():
        net.finish_wgrad(None)
        net.finish_grads()
    with torch.cuda.stream(cs):
        for _ in range(2):
            st = f(); b(st); w()
    torch.cuda.synchronize()
    tok = object()
    net._capture_token = tok
    gs = []
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=cs):
        st = f()
    gs.append(g)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=cs):
        b(st)
    gs.append(g)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=cs):
        w()
    gs.append(g)
    net._capture_token = None
    return net, st, gs


def bench(g, reps=30):
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


k32 = phases(32)
k64 = phases(64)
t_spin = time.perf_counter()
while time.perf_counter() - t_spin < 5.0:
    bench(k32[2][0], 10)
for rnd in range(2):
    for N, k in ((32, k32), (64, k64)):
        f_, b_, w_ = (bench(g) for g in k[2])
        print(f"N = {N}: forward {f_:.3f} ms | gradient chain {b_:.3f} | grouped weight gradients {w_:.3f}   (alone on the chip, linear graphs)", flush=True)
