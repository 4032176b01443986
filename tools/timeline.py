"""Dev tool: where a conv work-group's time goes (udapose_debug_set_timeline stamps, 10 ns ticks)."""
import sys
sys.path.insert(0, '.')
import torch
from uda_poseestimation_amd import ops, _hip
lib = _hip.lib()
N = 32
CASES = [("l3.c1 1024->256 1x1", 16, 1024, 256, 1, 0), ("l3.c2 256->256 3x3", 16, 256, 256, 3, 1), ("l3.c3 256->1024 1x1", 16, 256, 1024, 1, 0),
         ("l2.c2 128->128 3x3", 32, 128, 128, 3, 1), ("l1.c3 64->256 1x1", 64, 64, 256, 1, 0)]
buf = torch.zeros(1 << 16, 8, dtype=torch.int64, device='cuda')
for name, H, Ci, Co, K, pad in CASES:
    d = ops.conv_desc(N, H, H, Ci, Co, K, 1, pad)
    x = torch.randn(N, H, H, Ci, device='cuda').bfloat16()
    w = torch.randn(Co, K * K, Ci, device='cuda').bfloat16()
    for _ in range(3):
        ops.conv2d_fwd(x, w, d, want_stats=True)
    torch.cuda.synchronize()
    buf.zero_()
    lib.udapose_debug_set_timeline(buf.data_ptr())
    ops.conv2d_fwd(x, w, d, want_stats=True)
    torch.cuda.synchronize()
    lib.udapose_debug_set_timeline(None)
    t = buf.cpu()
    t = t[t[:, 0] > 0].double()
    nb = t.shape[0]
    t0 = t[:, 0].min()
    seg = [(t[:, i + 1] - t[:, i]).mean().item() / 100 for i in range(5)]
    print(f"{name:22s} blocks={nb:5d} kernel(span)={(t[:, 5].max() - t0).item() / 100:6.2f}us  start-spread={(t[:, 0].max() - t0).item() / 100:5.2f}us | "
          f"prologue {seg[0]:.2f}  first-stage {seg[1]:.2f}  k-loop {seg[2]:.2f}  epilogue {seg[3]:.2f}  drain {seg[4]:.2f} us (means per work-group)")
