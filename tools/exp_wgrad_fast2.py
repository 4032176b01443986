"""Dev experiment: the two fast-geometry loaders of the weight-gradient kernel, one layer at a time with the chip filled."""
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
for (N, H, Ci, Co, K, ks) in ((512, 16, 256, 256, 3, 32), (2048, 8, 512, 512, 3, 16), (512, 16, 1024, 256, 1, 32), (512, 16, 256, 1024, 1, 32), (128, 32, 128, 128, 3, 32), (32, 64, 64, 64, 3, 32)):
    d = ops.conv_desc(N, H, H, Ci, Co, K, 1, K // 2)
    x = torch.randn(N, H, H, Ci, device='cuda').bfloat16()
    dy = torch.randn(N, H, H, Co, device='cuda').bfloat16()
    flops = 2.0 * N * H * H * Ci * Co * K * K
    r = []
    for fg in (1, 2, 1, 2):
        dt = ops.with_policy(d, _hip.policy(wgrad_ksplit=ks, wgrad_fastgeo=fg, wgrad_row3=0))
        r.append(timeit(lambda: ops.conv2d_bwd_weight(dy, x, dt)))
    print(f"N={N} {H}x{H} {Ci}->{Co} k{K} ksplit {ks}: pointer loader {r[0]:7.1f} / {r[2]:7.1f} us ({flops / min(r[0], r[2]) / 1e6:5.0f} TFLOP/s)   buffer loader {r[1]:7.1f} / {r[3]:7.1f} us ({flops / min(r[1], r[3]) / 1e6:5.0f} TFLOP/s)", flush=True)
