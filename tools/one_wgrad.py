"""Dev tool: ONE weight-gradient layer launched a few times (for rocprofv3 --pmc runs).
usage: one_wgrad.py N H Ci Co K ksplit [tile] [row3]"""
import sys
sys.path.insert(0, '.')
import torch
from uda_poseestimation_amd import ops, _hip
N, H, Ci, Co, K, ks = (int(v) for v in sys.argv[1:7])
tile = int(sys.argv[7]) if len(sys.argv) > 7 else -1
row3 = int(sys.argv[8]) if len(sys.argv) > 8 else 0
d = ops.with_policy(ops.conv_desc(N, H, H, Ci, Co, K, 1, K // 2), _hip.policy(wgrad_ksplit=ks, wgrad_row3=row3, wgrad_tile=tile))
x = torch.randn(N, H, H, Ci, device='cuda').bfloat16()
dy = torch.randn(N, H, H, Co, device='cuda').bfloat16()
for _ in range(5):
    ops.conv2d_bwd_weight(dy, x, d)
torch.cuda.synchronize()
