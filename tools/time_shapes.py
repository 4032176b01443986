"""Dev tool: time the multi-tap PoseResNet-101 convolution shapes (N=32) with the heuristic tile, fprop and dgrad.
usage: [UDAPOSE_IGEMM_CMAJOR=1] python tools/time_shapes.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uda_poseestimation_amd import ops
N = 32
SHAPES = [("l1.c2 3x3 64", 64, 64, 64, 3, 1, 1, 0), ("l2.c2 3x3 128", 32, 128, 128, 3, 1, 1, 0), ("l2.c2s2 3x3", 64, 128, 128, 3, 2, 1, 0),
          ("l3.c2 3x3 256", 16, 256, 256, 3, 1, 1, 0), ("l4.c2 3x3 512", 8, 512, 512, 3, 1, 1, 0), ("up0 2048->256", 8, 2048, 256, 4, 2, 1, 1),
          ("up1 256->256", 16, 256, 256, 4, 2, 1, 1), ("up2 256->256", 32, 256, 256, 4, 2, 1, 1), ("l3.c1 1024->256", 16, 1024, 256, 1, 1, 0, 0)]
def timeit(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for name, H, Ci, Co, K, s, p, tr in SHAPES:
    d = ops.conv_desc(N, H, H, Ci, Co, K, s, p, transposed=bool(tr))
    ho, wo = ops.conv_out_hw(d)
    x = torch.randn(N, H, H, Ci, device='cuda').bfloat16()
    wshape = (Ci, Co, K, K) if tr else (Co, Ci, K, K)
    w = torch.randn(wshape, device='cuda') * 0.05
    wf, wb = ops.pack_weight(w, d, "fwd"), ops.pack_weight(w, d, "bwd")
    dy = torch.randn(N, ho, wo, Co, device='cuda').bfloat16()
    flops = 2.0 * N * (H * H if tr else ho * wo) * Co * Ci * K * K
    y = torch.empty(N, ho, wo, Co, device='cuda', dtype=torch.bfloat16)
    uf = timeit(lambda: ops.conv2d_fwd(x, wf, d, want_stats=True))
    ub = timeit(lambda: ops.conv2d_bwd_data(dy, wb, d))
    print(f"{name:20s} GF={flops/1e9:6.2f}  fprop {uf:6.1f} us {flops/uf/1e6:5.0f} TF   dgrad {ub:6.1f} us {flops/ub/1e6:5.0f} TF")
