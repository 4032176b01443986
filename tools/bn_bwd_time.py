"""Dev tool: time the pre-reduced BatchNorm backward (udapose_bn_bwd_pre) at the N=32 layer sizes with HIP events.
usage: python tools/bn_bwd_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uda_poseestimation_amd import ops

CASES = [("up_bn2 f32 131072x256", 131072, 256, True, 1024), ("up_bn1 f32 32768x256", 32768, 256, True, 256),
         ("l1.b3 131072x256", 131072, 256, False, 1024), ("l1.b1 131072x64", 131072, 64, False, 1024),
         ("l2.b1 32768x128", 32768, 128, False, 256), ("l2.b3 32768x512", 32768, 512, False, 256),
         ("l3.b3 8192x1024", 8192, 1024, False, 64), ("l3.b1 8192x256", 8192, 256, False, 128)]
for name, npix, C, f32, rows in CASES:
    g = torch.randn(npix, C, device="cuda", dtype=torch.float32 if f32 else torch.bfloat16)
    y = torch.randn(npix, C, device="cuda").bfloat16()
    gamma, mean, invstd = (torch.rand(C, device="cuda") + 0.5 for _ in range(3))
    slab = torch.randn(rows, 2, C, device="cuda")
    for _ in range(3):
        ops.bn_bwd_pre(g, y, gamma, mean, invstd, slab)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.bn_bwd_pre(g, y, gamma, mean, invstd, slab)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    nbytes = npix * C * ((4 if f32 else 2) + 2 + 2)
    print(f"{name:28s} {us:8.1f} us   {nbytes / us / 1e6:6.2f} TB/s (incl. allocation / finalize launches)")
