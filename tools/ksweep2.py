"""Dev tool: run a few 1x1 conv shapes (for rocprofv3 --kernel-trace); prints the launch order."""
import sys
sys.path.insert(0, '.')
import torch
from uda_poseestimation_amd import ops, _hip
lib = _hip.lib()
N = 32
for Co, t in ((1024, 6), (256, 2), (256, 6)):
    lib.udapose_debug_set_tiles(t, -1, -1)
    for Ci in (64, 256, 1024, 2048):
        d = ops.conv_desc(N, 16, 16, Ci, Co, 1)
        x = torch.randn(N, 16, 16, Ci, device='cuda').bfloat16()
        w = torch.randn(Co, 1, Ci, device='cuda').bfloat16()
        for st in (True, False):
            for _ in range(10):
                ops.conv2d_fwd(x, w, d, want_stats=st)
            torch.cuda.synchronize()
            print(f"Co={Co} tile={t} Ci={Ci} stats={st}")
