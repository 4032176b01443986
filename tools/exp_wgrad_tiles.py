"""Dev experiment: is the weight-gradient kernel bound by its LDS fill?  One 3x3 256->256 layer at 16x16 with enough images to fill
the chip by itself, 128x128 tiles (64 FLOP per filled byte) against 64x64 tiles (32 FLOP per byte), same work-group count."""
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
for (N, H, C) in ((512, 16, 256), (128, 32, 128), (2048, 8, 512)):
    d = ops.conv_desc(N, H, H, C, C, 3, 1, 1)
    x = torch.randn(N, H, H, C, device='cuda').bfloat16()
    dy = torch.randn(N, H, H, C, device='cuda').bfloat16()
    flops = 2.0 * N * H * H * C * C * 9
    for t, ks in ((0, 8), (0, 16), (0, 32), (1, 2), (1, 4), (1, 8)):
        dt = ops.with_policy(d, _hip.policy(wgrad_tile=t, wgrad_ksplit=ks))
        us = timeit(lambda: ops.conv2d_bwd_weight(dy, x, dt))
        tiles = (C // (128 if t == 0 else 64)) ** 2 * 9 * ks
        print(f"N={N} {H}x{H} C={C}: tile {'128x128' if t == 0 else '64x64  '} ksplit {ks:2d} ({tiles:5d} work-groups): {us:8.1f} us = {flops / us / 1e6:6.0f} TFLOP/s", flush=True)
