"""Dev tool: time the AdaIN style pass (Style_net.Net.forward) at the benchmark batch."""
import sys, time
sys.path.insert(0, '.')
import torch
from uda_poseestimation_amd.lib.models import Style_net
from uda_poseestimation_amd import synthetic
sys.path.insert(0, 'tests/golden')
from seeded import fill_style_weights
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
fill_style_weights(Style_net.vgg, 1); fill_style_weights(Style_net.decoder, 2)
net = Style_net.Net(Style_net.vgg, Style_net.decoder).cuda()
c = synthetic.images(N, 256, 1).cuda(); s = synthetic.images(N, 256, 2).cuda()
with torch.no_grad():
    for _ in range(3):
        out = net(c, s, 0.5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        out = net(c, s, 0.5)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
print(f"style pass N={N}: {dt * 1e3:.1f} ms  ({N * 94.9e9 / dt / 1e12:.0f} TFLOP/s of the minimal 94.9 GFLOP per pair)")
