"""Dev tool: layer3's three forward convolutions of PoseResNet-101 (N = 32) through the product's igemm, each replayed back to back from a hipGraph
(no host launch cost in the figure) - the counterpart of the 'conv ... alone' rows of tools/probe/conv_bn_seam.hip.  usage: python tools/time_l3_convs.py"""
import sys
sys.path.insert(0, '.')
import torch
from uda_poseestimation_amd import ops, _hip
N, REP = 32, 69
SHAPES = [("c1 1024->256", 16, 1024, 256, 1, 0), ("c2 3x3 256", 16, 256, 256, 3, 1), ("c3 256->1024", 16, 256, 1024, 1, 0),
          ("l2 c1 512->128", 32, 512, 128, 1, 0), ("l2 c2 3x3 128", 32, 128, 128, 3, 1), ("l2 c3 128->512", 32, 128, 512, 1, 0),
          ("l4 c1 2048->512", 8, 2048, 512, 1, 0), ("l4 c3 512->2048", 8, 512, 2048, 1, 0)]
for name, H, Ci, Co, K, p in SHAPES:
    x = torch.randn(N, H, H, Ci, device='cuda').bfloat16()
    w = (torch.randn(Co, K * K, Ci, device='cuda') * 0.05).bfloat16()
    row = []
    for t in (-1, 0, 4, 1, 6, 2, 5):
        d = ops.conv_desc(N, H, H, Ci, Co, K, 1, p)
        if t >= 0:
            d = ops.with_policy(d, _hip.policy(igemm_tile=t))
        try:
            ops.conv2d_fwd(x, w, d, want_stats=True)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                with torch.cuda.graph(g, stream=s):
                    for _ in range(REP):
                        keep = ops.conv2d_fwd(x, w, d, want_stats=True)
            best = 1e9
            for _ in range(5):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); g.replay(); b.record(); torch.cuda.synchronize()
                best = min(best, a.elapsed_time(b) * 1e3 / REP)
            row.append(f"{'auto' if t < 0 else 't%d' % t}: {best:5.2f}")
        except Exception as e:
            row.append(f"t{t}: fail")
    print(f"{name:18s} us per launch | " + " | ".join(row), flush=True)
