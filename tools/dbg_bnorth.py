import sys
sys.path.insert(0, '.')
import torch
from uda_poseestimation_amd import ops
N, H, C_ = 32, 16, 1024
g = torch.Generator(device="cuda").manual_seed(C_)
y = torch.randn(N, H, H, C_, device="cuda", generator=g).bfloat16()
yf = y.float()
mean = yf.mean((0, 1, 2)); invstd = 1.0 / torch.sqrt(yf.var((0, 1, 2), unbiased=False) + 1e-5)
gamma = torch.rand(C_, device="cuda", generator=g) + 0.5
beta = torch.randn(C_, device="cuda", generator=g) * 0.1
dz = torch.randn(N, H, H, C_, device="cuda", generator=g)
dy, dgamma, dbeta, gm = ops.bn_bwd(dz, None, y, gamma, mean, invstd, relu=2, want_g=True, beta=beta)
xh = ((yf - mean) * invstd).double()
M = N * H * H
sc = gamma * invstd; sh = beta - mean * sc
gd = dz.double() * ((yf * sc + sh) > 0)
ideal = (gamma * invstd).double() * (gd - gd.sum((0, 1, 2)) / M - xh * (gd * xh).sum((0, 1, 2)) / M)
print("sum ideal max", float(ideal.sum((0, 1, 2)).abs().max()))
print("sum dev   max", float(dy.double().sum((0, 1, 2)).abs().max()))
err = dy.double() - ideal
print("err mean per channel max", float(err.mean((0, 1, 2)).abs().max()), "err rms", float(err.pow(2).mean().sqrt()), "dy rms", float(ideal.pow(2).mean().sqrt()))
r = ideal.float().bfloat16().double() - ideal
print("pure bf16 rounding: sum max", float(r.sum((0, 1, 2)).abs().max()), "rms", float(r.pow(2).mean().sqrt()))
print("gout vs g", float((gm.double() - gd.float().bfloat16().double()).abs().max()))
