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


# ---- generalisation (round 6): any 16-bit storage type, rounding switched per (stage, tensor kind) - tools/attribute_16bit_error.py
STAGES = ("stem", "layer1", "layer2", "layer3", "layer4", "up0", "up1", "up2", "head")


def forward_emulated(m, x, dtype, sel):
    """m: PoseResNetRef in train mode; `dtype` torch.float16 / torch.bfloat16; sel(stage, kind) -> bool says whether the tensors of that kind
    are stored in `dtype` in that stage: kind "x" (the input image, stage "stem"), "w" (weight packs), "y" (pre-BatchNorm convolution outputs),
    "z" (post-BatchNorm / ReLU outputs).  sel = always True reproduces forward_bf16_emulated's storage points (BatchNorm statistics from the
    un-rounded convolution result, as the executor's fused epilogue computes them)."""
    def r(t, st, kind):
        return t.to(dtype).float() if sel(st, kind) else t

    def bn(y32, b, st, res=None, relu=True):
        mean = y32.mean((0, 2, 3))
        var = y32.var((0, 2, 3), unbiased=False)
        scale = b.weight / torch.sqrt(var + b.eps)
        shift = b.bias - mean * scale
        z = r(y32, st, "y") * scale[None, :, None, None] + shift[None, :, None, None]
        if res is not None:
            z = z + res
        return r(F.relu(z) if relu else z, st, "z")

    b = m.backbone
    z = bn(F.conv2d(r(x, "stem", "x"), r(b.conv1.weight, "stem", "w"), stride=2, padding=3), b.bn1, "stem")
    z = F.max_pool2d(z, 3, 2, 1)
    for li, layer in enumerate((b.layer1, b.layer2, b.layer3, b.layer4)):
        st = f"layer{li + 1}"
        for blk in layer:
            idt = z
            z1 = bn(F.conv2d(z, r(blk.conv1.weight, st, "w")), blk.bn1, st)
            z2 = bn(F.conv2d(z1, r(blk.conv2.weight, st, "w"), stride=blk.conv2.stride, padding=1), blk.bn2, st)
            if blk.downsample is not None:
                idt = bn(F.conv2d(z, r(blk.downsample[0].weight, st, "w"), stride=blk.downsample[0].stride), blk.downsample[1], st, relu=False)
            z = bn(F.conv2d(z2, r(blk.conv3.weight, st, "w")), blk.bn3, st, res=idt)
    up = m.upsampling
    for k, i in enumerate((0, 3, 6)):
        z = bn(F.conv_transpose2d(z, r(up[i].weight, f"up{k}", "w"), stride=2, padding=1), up[i + 1], f"up{k}")
    return F.conv2d(z, r(m.head.weight, "head", "w"), m.head.bias)
