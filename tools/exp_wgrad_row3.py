"""Dev experiment: the filter-row weight-gradient form against the one-tap forms, one 3x3 layer at a time with enough images to
fill the chip by itself."""
import sys
sys.path.insert(0, '.')
import torch
from uda_poseestimation_amd import ops, _hip
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for (N, H, C) in ((512, 16, 256), (128, 32, 128), (2048, 8, 512), (32, 64, 64)):
    d = ops.conv_desc(N, H, H, C, C, 3, 1, 1)
    x = torch.randn(N, H, H, C, device='cuda').bfloat16()
    dy = torch.randn(N, H, H, C, device='cuda').bfloat16()
    flops = 2.0 * N * H * H * C * C * 9
    for name, pol in (("one tap, heuristic tile", dict(wgrad_row3=0)), ("filter row 64x64      ", dict(wgrad_row3=1))):
        for ks in (4, 8, 16, 32):
            dt = ops.with_policy(d, _hip.policy(wgrad_ksplit=ks, **pol))
            us = timeit(lambda: ops.conv2d_bwd_weight(dy, x, dt))
            print(f"N={N} {H}x{H} C={C}: {name} ksplit {ks:2d}: {us:8.1f} us = {flops / us / 1e6:6.0f} TFLOP/s", flush=True)
