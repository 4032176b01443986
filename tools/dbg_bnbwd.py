"""Dev tool: BN backward kernel (fp32 dz, both mask modes) against torch on one layer shape."""
import sys
sys.path.insert(0, '.')
import torch
from uda_poseestimation_amd import ops
torch.manual_seed(0)
for (N, H, C_, f32) in ((4, 32, 256, True), (4, 32, 256, False), (8, 16, 2048, True), (32, 16, 1024, False), (4, 16, 64, False)):
    y = torch.randn(N, H, H, C_, device='cuda').bfloat16()
    gamma = (torch.rand(C_, device='cuda') + 0.5); beta = torch.randn(C_, device='cuda') * 0.1
    yf = y.float()
    mean = yf.mean((0, 1, 2)); var = yf.var((0, 1, 2), unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    sc = gamma * invstd; sh = beta - mean * sc
    z = torch.relu(yf * sc + sh).bfloat16()
    dz = torch.randn(N, H, H, C_, device='cuda')
    dzk = dz if f32 else dz.bfloat16()
    g = dzk.float() * (z.float() > 0)
    xh = (yf - mean) * invstd
    M = N * H * H
    dbeta = g.sum((0, 1, 2)); dgamma = (g * xh).sum((0, 1, 2))
    dy = sc * (g - dbeta / M - xh * dgamma / M)
    for mode in (1, 2):
        o = ops.bn_bwd(dzk, z if mode == 1 else None, y, gamma, mean, invstd, relu=mode, want_g=True, beta=beta)
        print(N, H, C_, "f32" if f32 else "bf16", "mode", mode, "dy", float((o[0].float() - dy).abs().max()), "dgamma", float((o[1] - dgamma).abs().max() / dgamma.abs().max()),
              "dbeta", float((o[2] - dbeta).abs().max() / dbeta.abs().max()), "g", float((o[3].float() - g).abs().max()))
