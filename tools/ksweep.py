"""Dev tool: fixed overhead vs per-K-step cost of the igemm kernel (1x1 conv, 16x16 maps)."""
import sys
sys.path.insert(0, '.')
import torch
from uda_poseestimation_amd import ops, _hip
lib = _hip.lib()
def timeit(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for N in (32, 64):
    for Co in (256, 1024):
        for t in (2, 5, 6):
            lib.udapose_debug_set_tiles(t, -1, -1)
            row = []
            for Ci in (64, 128, 256, 512, 1024, 2048, 4096):
                d = ops.conv_desc(N, 16, 16, Ci, Co, 1)
                x = torch.randn(N, 16, 16, Ci, device='cuda').bfloat16()
                w = torch.randn(Co, 1, Ci, device='cuda').bfloat16()
                us = timeit(lambda: ops.conv2d_fwd(x, w, d, want_stats=True))
                row.append(us)
            print(f"N={N} Co={Co} tile={t}: " + " ".join(f"K{k}:{u:6.1f}" for k, u in zip((64, 128, 256, 512, 1024, 2048, 4096), row)), flush=True)
