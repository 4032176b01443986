import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests/golden')
import numpy as np, torch
from uda_poseestimation_amd.utils import OldWeightEMA
z=np.load('tests/golden/ema.npz')
class Holder(torch.nn.Module):
    def __init__(self, arrs):
        super().__init__()
        self.ps = torch.nn.ParameterList([torch.nn.Parameter(torch.from_numpy(a).clone()) for a in arrs])
stu = Holder([z[f"src{i}"] for i in range(3)]).cuda()
tea = Holder([np.zeros_like(z[f"src{i}"]) for i in range(3)]).cuda()
ema = OldWeightEMA(tea, stu, alpha=0.999)
for i,p in enumerate(tea.parameters()): print("init eq", np.array_equal(p.detach().cpu().numpy(), z[f"init{i}"]))
cpu_t=[torch.from_numpy(z[f"init{i}"]).clone() for i in range(3)]
for it in range(3):
    with torch.no_grad():
        for i,p in enumerate(stu.parameters()): p.copy_(torch.from_numpy(z[f"stu_it{it}_{i}"]))
    ema.step()
    for i in range(3):
        s=torch.from_numpy(z[f"stu_it{it}_{i}"])
        cpu_t[i].mul_(0.999); cpu_t[i].add_(s*(1.0-0.999))
    for i,p in enumerate(tea.parameters()):
        d=(p.detach().cpu()-cpu_t[i]).abs().max().item()
        print(it,i,"dev vs cpu recompute maxdiff",d)
for i,p in enumerate(tea.parameters()):
    print("final", np.array_equal(p.detach().cpu().numpy(), z[f"final{i}"]), np.abs(p.detach().cpu().numpy()-z[f"final{i}"]).max(), np.array_equal(cpu_t[i].numpy(), z[f"final{i}"]))
