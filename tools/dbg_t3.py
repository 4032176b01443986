import torch, sys
sys.path.insert(0,'.')
from uda_poseestimation_amd import ops
import torch.nn.functional as F
g = torch.Generator().manual_seed(1)
N,H,W,Ci,Co=2,16,16,256,32
x = torch.randn(N,Ci,H,W,generator=g).bfloat16().float()
dy = torch.randn(N,Co,H,W,generator=g).bfloat16().float()
d = ops.conv_desc(N,H,W,Ci,Co,1)
nh = lambda t: t.permute(0,2,3,1).contiguous().bfloat16().cuda()
dw = ops.conv2d_bwd_weight(nh(dy), nh(x), d)
torch.cuda.synchronize()
ref = torch.einsum('nohw,nchw->oc', dy, x)
print('tile3 err', (dw.cpu().reshape(Co,Ci)-ref).abs().max().item(), ref.abs().max().item())
