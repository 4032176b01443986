"""Dev check: Policy::bn_fin_inkernel (statistics finished inside the convolution launches) against the finalize-launch form on the same weights and input:
forward outputs, running statistics, and - with a backward - every parameter gradient.  usage: python tools/check_fin.py [N]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import uda_poseestimation_amd.lib.models as models

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.randn(N, 3, 256, 256, device=dev)
d = torch.randn(N, 16, 64, 64, device=dev)
res = {}
for fin in (0, 1):
    torch.manual_seed(1)
    net = models.pose_resnet101(16, pretrained_backbone=False).to(dev)
    net.precision = "bf16"
    net.policy.update({"bn_fin_inkernel": fin})
    net.train()
    outs = []
    for it in range(3):              # (several passes: the arrival counters must have reset themselves)
        for p in net.parameters():
            p.grad = None
        y = net(x)
        y.backward(d)
        outs.append(y.detach().float().clone())
    torch.cuda.synchronize()
    res[fin] = (outs, {k: v.detach().clone() for k, v in net.state_dict().items()}, {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None})
o0, s0, g0 = res[0]
o1, s1, g1 = res[1]
for it in range(3):
    print(f"pass {it}: max |dy| {float((o0[it] - o1[it]).abs().max()):.3e} of max |y| {float(o0[it].abs().max()):.3e}, finite {bool(torch.isfinite(o1[it]).all())}")
worst = max(((float((s0[k].float() - s1[k].float()).abs().max()) / (float(s0[k].float().abs().max()) + 1e-30), k) for k in s0), key=lambda t: t[0])
print("state_dict (weights + running statistics): worst relative difference", worst)
worstg = max(((float((g0[k] - g1[k]).abs().max()) / (float(g0[k].abs().max()) + 1e-30), k) for k in g0), key=lambda t: t[0])
print("gradients: worst relative difference", worstg)
