"""Dev tool: time every distinct PoseResNet-101 convolution shape (N=32, 256x256) under each tile configuration."""
import sys, itertools
sys.path.insert(0, '.')
import torch
from uda_poseestimation_amd import ops, _hip
lib = _hip.lib()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
which = sys.argv[2] if len(sys.argv) > 2 else "fwd"
# (name, H, Ci, Co, K, stride, pad, transposed, count per forward)
SHAPES = [
    ("l1.c1 64->64 1x1", 64, 64, 64, 1, 1, 0, 0, 1), ("l1.c1 256->64", 64, 256, 64, 1, 1, 0, 0, 2),
    ("l1.c2 3x3 64", 64, 64, 64, 3, 1, 1, 0, 3), ("l1.c3 64->256", 64, 64, 256, 1, 1, 0, 0, 4),
    ("l2.c1 512->128", 32, 512, 128, 1, 1, 0, 0, 3), ("l2.c2 3x3 128", 32, 128, 128, 3, 1, 1, 0, 3),
    ("l2.c3 128->512", 32, 128, 512, 1, 1, 0, 0, 4), ("l2.c2s2 3x3", 64, 128, 128, 3, 2, 1, 0, 1),
    ("l3.c1 1024->256", 16, 1024, 256, 1, 1, 0, 0, 22), ("l3.c2 3x3 256", 16, 256, 256, 3, 1, 1, 0, 22),
    ("l3.c3 256->1024", 16, 256, 1024, 1, 1, 0, 0, 23), ("l3.ds 512->1024 s2", 32, 512, 1024, 1, 2, 0, 0, 1),
    ("l4.c1 2048->512", 8, 2048, 512, 1, 1, 0, 0, 2), ("l4.c2 3x3 512", 8, 512, 512, 3, 1, 1, 0, 2),
    ("l4.c3 512->2048", 8, 512, 2048, 1, 1, 0, 0, 3), ("up0 2048->256", 8, 2048, 256, 4, 2, 1, 1, 1),
    ("up1 256->256", 16, 256, 256, 4, 2, 1, 1, 1), ("up2 256->256", 32, 256, 256, 4, 2, 1, 1, 1),
]
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
tot = {}
for name, H, Ci, Co, K, s, p, tr, cnt in SHAPES:
    d = ops.conv_desc(N, H, H, Ci, Co, K, s, p, transposed=bool(tr))
    ho, wo = ops.conv_out_hw(d)
    x = torch.randn(N, H, H, Ci, device='cuda').bfloat16()
    flops = 2.0 * N * (H * H if tr else ho * wo) * Co * Ci * K * K
    if which == "fwd":
        w = torch.randn(Co, K * K, Ci, device='cuda').bfloat16()
        res = []
        for t in (0, 4, 1, 6, 2, 5, 7, 8):
            dt = ops.with_policy(d, _hip.policy(igemm_tile=t))       # the tile is part of the call's explicit dispatch policy
            try:
                us = timeit(lambda: ops.conv2d_fwd(x, w, dt, want_stats=True))
            except Exception as e:
                us = float('nan')
            res.append((t, us))
        best = min(res, key=lambda r: r[1] if r[1] == r[1] else 1e9)
        tot[name] = best[1] * cnt
        print(f"{name:22s} GF={flops/1e9:6.2f} " + " ".join(f"t{t}:{us:6.1f}" for t, us in res) + f"  best t{best[0]} {flops/best[1]/1e6:6.0f} TF")
    else:
        dy = torch.randn(N, ho, wo, Co, device='cuda').bfloat16()
        res = []
        for t, ks in itertools.product((0, 1), (1, 2, 4, 8, 16, 32, 64)):
            dt = ops.with_policy(d, _hip.policy(wgrad_tile=t, wgrad_ksplit=ks))
            try:
                us = timeit(lambda: ops.conv2d_bwd_weight(dy, x, dt))
            except Exception as e:
                us = float('nan')
            res.append((t, ks, us))
        best = min(res, key=lambda r: r[2] if r[2] == r[2] else 1e9)
        tot[name] = best[2] * cnt
        print(f"{name:22s} GF={flops/1e9:6.2f} " + " ".join(f"t{t}k{ks}:{us:5.0f}" for t, ks, us in res) + f"  best t{best[0]}k{best[1]} {flops/best[2]/1e6:6.0f} TF")
print("sum of best x count (us per forward-equivalent):", sum(tot.values()))
