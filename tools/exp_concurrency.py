"""Dev experiment: do two independent forward+backward passes overlap usefully on two HIP streams?"""
import sys, time
sys.path.insert(0, '.')
import torch
import uda_poseestimation_amd.lib.models as models
torch.manual_seed(0)
a = models.pose_resnet101(16, pretrained_backbone=False).cuda()
b = models.pose_resnet101(16, pretrained_backbone=False).cuda()
x1 = torch.randn(32, 3, 256, 256, device='cuda'); x2 = torch.randn(32, 3, 256, 256, device='cuda')
s2 = torch.cuda.Stream()
def seq():
    a(x1).square().mean().backward(); b(x2).square().mean().backward()
def par():
    main = torch.cuda.current_stream()
    s2.wait_stream(main)
    with torch.cuda.stream(s2):
        b(x2).square().mean().backward()
    a(x1).square().mean().backward()
    main.wait_stream(s2)
for fn in (seq, par, seq, par):
    for _ in range(25): fn()      # includes clock spin-up
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize(); print(fn.__name__, (time.perf_counter() - t0) / 10 * 1e3, "ms")
