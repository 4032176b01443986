"""Dev tool: time the re-warp kernel (forward and backward) at the benched heat-map size."""
import sys
sys.path.insert(0, '.')
import torch
from uda_poseestimation_amd import warp, synthetic
import numpy as np
N = 32
ap = synthetic.aug_params(N, np.random.RandomState(3))
th = warp.recon_thetas(ap, N, 4.0, "cuda")
y = torch.randn(N, 16, 64, 64, device="cuda", requires_grad=True)
g = torch.randn(N, 16, 64, 64, device="cuda")
def t(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
fwd = lambda: warp.warp_chain(y.detach(), th)
def fb():
    y.grad = None
    warp.warp_chain(y, th).backward(g)
tf = t(fwd); tfb = t(fb)
print(f"warp_chain forward {tf:.1f} us; forward + backward (incl. autograd host work) {tfb:.1f} us")
