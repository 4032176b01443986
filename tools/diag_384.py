"""Dev tool: which forward path differs at 384x384 (BASELINE.json configs[4])?  The 16-bit forward under single policy switches
against the f16x2 forward (fp32-grade) of the same network."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import uda_poseestimation_amd.lib.models.pose_resnet as pr

def run(S, N, layers, K=18, prec="fp16", pol=None, seed=21):
    torch.manual_seed(seed)
    net = pr._pose_resnet("d", K, pr.Bottleneck_default, layers, False, False)
    g = torch.Generator().manual_seed(22)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) * 0.5 + 0.75)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
            if hasattr(m, "bn3"):
                m.bn3.weight.fill_(0.1)
    net = net.cuda().train()
    x = torch.randn(N, 3, S, S, generator=g).cuda()
    net.precision = "f16x2"
    with torch.no_grad():
        y0 = net(x)
    net.precision = prec
    net.policy = dict(pol or {})
    net._handles = {}
    with torch.no_grad():
        y = net(x)
    return (y - y0).abs().max().item(), y0.abs().max().item()

for S, N, layers in ((384, 2, [3, 4, 23, 3]), (256, 2, [3, 4, 23, 3]), (384, 2, [1, 1, 1, 1]), (384, 4, [3, 4, 23, 3]), (320, 2, [3, 4, 23, 3])):
    for prec in ("fp16", "bf16"):
        e, s = run(S, N, layers, prec=prec)
        print(f"S={S} N={N} layers={layers} {prec}: max|y - f16x2| = {e:.3e} (max|y| {s:.3f})", flush=True)
for pol in ({"igemm_h3": 0}, {"igemm_lean": 0}, {"bn_fwd_chunked": 0}, {"stem_fused": 0}, {"igemm_short_lds": 0}, {"igemm_tap0": 0}, {"bn3_mask": 0},
            {"igemm_wg_min": 100000}, {"igemm_tile": 0}):
    e, s = run(384, 2, [3, 4, 23, 3], pol=pol)
    print(f"S=384 N=2 R101 fp16 policy {pol}: {e:.3e}", flush=True)
