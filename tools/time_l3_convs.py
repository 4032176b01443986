"""Dev tool: the forward convolutions of PoseResNet-101's bottlenecks (N = 32, 256x256) through the product's igemm under every tile id, each replayed back
to back from a hipGraph (no host launch cost in the figure; tools/tune_conv.py's eager launches are host-bound below ~12.6 us) - also the counterpart of
the 'conv ... alone' rows of tools/probe/conv_bn_seam.hip.  Tile ids: igemm.hip, igemm_pick_tile (10 / 11 / 12: the run-staged 3x3 forms, which fall back
to plain tiles on other shapes).  usage: python tools/time_l3_convs.py [name prefix, e.g. l3]"""
import sys
sys.path.insert(0, '.')
import torch
from uda_poseestimation_amd import ops, _hip
N, REP = 32, 69
SHAPES = [("l1 c1 64->64", 64, 64, 64, 1, 0), ("l1 c1 256->64", 64, 256, 64, 1, 0), ("l1 c2 3x3 64", 64, 64, 64, 3, 1), ("l1 c3 64->256", 64, 64, 256, 1, 0),
          ("l2 c1 512->128", 32, 512, 128, 1, 0), ("l2 c2 3x3 128", 32, 128, 128, 3, 1), ("l2 c3 128->512", 32, 128, 512, 1, 0),
          ("l3 c1 1024->256", 16, 1024, 256, 1, 0), ("l3 c2 3x3 256", 16, 256, 256, 3, 1), ("l3 c3 256->1024", 16, 256, 1024, 1, 0),
          ("l4 c1 2048->512", 8, 2048, 512, 1, 0), ("l4 c2 3x3 512", 8, 512, 512, 3, 1), ("l4 c3 512->2048", 8, 512, 2048, 1, 0)]
if len(sys.argv) > 1:
    SHAPES = [s for s in SHAPES if s[0].startswith(sys.argv[1])]
for name, H, Ci, Co, K, p in SHAPES:
    x = torch.randn(N, H, H, Ci, device='cuda').bfloat16()
    w = (torch.randn(Co, K * K, Ci, device='cuda') * 0.05).bfloat16()
    row = []
    for t in (-1, 4, 6, 9, 5, 10, 11):
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
