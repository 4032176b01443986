"""Dev tool: ONE weight-gradient layer launched a few times (for rocprofv3 --pmc runs).  usage: one_wgrad.py N H C ksplit [row3]"""
import sys
sys.path.insert(0, '.')
import torch
from uda_poseestimation_amd import ops, _hip
N, H, C, ks = (int(v) for v in sys.argv[1:5])
row3 = int(sys.argv[5]) if len(sys.argv) > 5 else 0
d = ops.with_policy(ops.conv_desc(N, H, H, C, C, 3, 1, 1), _hip.policy(wgrad_ksplit=ks, wgrad_row3=row3))
x = torch.randn(N, H, H, C, device='cuda').bfloat16()
dy = torch.randn(N, H, H, C, device='cuda').bfloat16()
for _ in range(5):
    ops.conv2d_bwd_weight(dy, x, d)
torch.cuda.synchronize()
