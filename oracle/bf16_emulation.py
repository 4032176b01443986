"""fp32 CPU evaluation of PoseResNetRef with bf16 rounding inserted at exactly the points where the MI355X executor
stores bf16 (conv inputs/weights, conv outputs, BN/ReLU outputs).  Test-only: it separates WIRING errors (which show
up at 1e-2 and above) from the bf16 storage noise that the plain fp32 oracle comparison has to tolerate.
Follows the same reference lines as oracle/pose_resnet_ref.py; BN statistics come from the un-rounded conv result,
as the executor's fused epilogue computes them (csrc/igemm.hip)."""
import torch
import torch.nn.functional as F


def q(t):
    """bf16 storage rounding; straight-through for autograd (the device backward differentiates the un-rounded maps)."""
    r = t.detach().to(torch.bfloat16).float()
    return t + (r - t.detach()) if t.requires_grad else r


def _bn_train(y32, bn, res=None, relu=True):
    mean = y32.mean((0, 2, 3))
    var = y32.var((0, 2, 3), unbiased=False)
    invstd = 1.0 / torch.sqrt(var + bn.eps)
    scale = bn.weight * invstd
    shift = bn.bias - mean * scale
    z = q(y32) * scale[None, :, None, None] + shift[None, :, None, None]
    if res is not None:
        z = z + res
    return q(F.relu(z) if relu else z)


def forward_bf16_emulated(m, x):
    """m: PoseResNetRef in train mode (batch statistics); returns fp32 heat-maps.  Differentiable when grad mode is on:
    the backward then sees the same stored (rounded) activations as the device backward."""
    if True:
        b = m.backbone
        x8 = q(x)
        z = _bn_train(F.conv2d(x8, q(b.conv1.weight), stride=2, padding=3), b.bn1)
        z = F.max_pool2d(z, 3, 2, 1)
        for layer in (b.layer1, b.layer2, b.layer3, b.layer4):
            for blk in layer:
                idt = z
                z1 = _bn_train(F.conv2d(z, q(blk.conv1.weight)), blk.bn1)
                z2 = _bn_train(F.conv2d(z1, q(blk.conv2.weight), stride=blk.conv2.stride, padding=1), blk.bn2)
                if blk.downsample is not None:
                    idt = _bn_train(F.conv2d(z, q(blk.downsample[0].weight), stride=blk.downsample[0].stride), blk.downsample[1],
                                    relu=False)
                z = _bn_train(F.conv2d(z2, q(blk.conv3.weight)), blk.bn3, res=idt)
        up = m.upsampling
        for i in (0, 3, 6):
            z = _bn_train(F.conv_transpose2d(z, q(up[i].weight), stride=2, padding=1), up[i + 1])
        return F.conv2d(z, q(m.head.weight), m.head.bias)
