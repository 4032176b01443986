"""Dev tool: run ONE convolution shape (fwd / dgrad / wgrad) repeatedly, for rocprofv3 --pmc runs.
usage: one_conv.py kind H Ci Co K stride pad transposed [igemm_tile wgrad_tile ksplit reps N]"""
import sys
sys.path.insert(0, '.')
import torch
from uda_poseestimation_amd import ops, _hip
a = sys.argv[1:]
kind, H, Ci, Co, K, s, p, tr = a[0], *[int(x) for x in a[1:8]]
it, wt, ks, reps, N = [int(x) for x in (a[8:13] + ["-1", "-1", "-1", "10", "32"][len(a[8:13]):])]
lib = _hip.lib()
lib.udapose_debug_set_tiles(it, wt, ks)
d = ops.conv_desc(N, H, H, Ci, Co, K, s, p, transposed=bool(tr))
ho, wo = ops.conv_out_hw(d)
x = torch.randn(N, H, H, Ci, device='cuda').bfloat16()
dy = torch.randn(N, ho, wo, Co, device='cuda').bfloat16()
T = K * K
w = torch.randn(Co, T, Ci, device='cuda').bfloat16()
wb = torch.randn(Ci, T, Co, device='cuda').bfloat16()
for _ in range(reps):
    if kind == "fwd": ops.conv2d_fwd(x, w, d, want_stats=True)
    elif kind == "dgrad": ops.conv2d_bwd_data(dy, wb, d)
    else: ops.conv2d_bwd_weight(dy, x, d)
torch.cuda.synchronize()
print("done")
