"""Dev experiment: 128x256 / 256x128 weight-gradient tiles (64x128 per wave) against 128x128 (64x64 per wave), per layer shape,
the chip filled by the split count; results checked against the 128x128 kernel's."""
import sys
sys.path.insert(0, '.')
import torch
from uda_poseestimation_amd import ops, _hip
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
# (N, H, Ci, Co, K)
for (N, H, Ci, Co, K) in ((512, 16, 256, 256, 3), (512, 16, 1024, 256, 1), (2048, 8, 512, 512, 3), (2048, 8, 2048, 512, 1)):
    d = ops.conv_desc(N, H, H, Ci, Co, K, 1, K // 2)
    x = torch.randn(N, H, H, Ci, device='cuda').bfloat16()
    dy = torch.randn(N, H, H, Co, device='cuda').bfloat16()
    flops = 2.0 * N * H * H * Ci * Co * K * K
    ref = None
    for t in (0, 5, 6):
        tr, tc = {0: (128, 128), 4: (128, 256), 5: (256, 128), 6: (256, 256)}[t]
        if Co % tr or Ci % tc:
            continue
        slots = {0: 512, 4: 512, 5: 512, 6: 256}[t]                  # resident work-groups of the per-layer kernels
        base = (Co // tr) * (Ci // tc) * K * K
        cands = sorted({max(1, (slots * r) // base) for r in (1, 2, 3)})
        for ks in cands:
            tiles = base * ks
            if (N * H * H) // (64 * ks) < 8:
                continue
            dt = ops.with_policy(d, _hip.policy(wgrad_tile=t, wgrad_ksplit=ks))
            dw = ops.conv2d_bwd_weight(dy, x, dt)
            if ref is None:
                ref = dw
            err = (dw - ref).abs().max().item() / ref.abs().max().item()
            us = timeit(lambda: ops.conv2d_bwd_weight(dy, x, dt))
            print(f"N={N} {H}x{H} {Ci}->{Co} k{K}: tile {tr}x{tc} ksplit {ks:2d} ({tiles:5d} WGs): {us:8.1f} us = {flops / us / 1e6:6.0f} TFLOP/s  err vs first {err:.1e}", flush=True)
