"""Dev tool: layer-by-layer comparison of the GPU kernels (chained from Python) with the bf16-emulating oracle."""
import sys
sys.path.insert(0, '.')
import torch, torch.nn.functional as F
from oracle.pose_resnet_ref import PoseResNetRef
from oracle.bf16_emulation import q, _bn_train
from uda_poseestimation_amd import ops
import uda_poseestimation_amd.lib.models.pose_resnet as pr

torch.manual_seed(0)
layers = [int(a) for a in (sys.argv[1] if len(sys.argv) > 1 else "1,1,1,1").split(",")]
N, HW = int(sys.argv[2]) if len(sys.argv) > 2 else 4, int(sys.argv[3]) if len(sys.argv) > 3 else 128
ref = PoseResNetRef(layers, 16).train()
x = torch.randn(N, 3, HW, HW)
nh = lambda t: t.permute(0, 2, 3, 1).contiguous().bfloat16().cuda()
nc = lambda t: t.float().cpu().permute(0, 3, 1, 2)


def rel(a, b):
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def gpu_conv_bn(xg, conv, bn, res=None, relu=True, transposed=False, smallc=False):
    Nn, H, W, Ci = xg.shape
    w = conv.weight.detach()
    if transposed:
        d = ops.conv_desc(Nn, H, W, Ci, w.shape[1], 4, 2, 1, transposed=True)
    else:
        d = ops.conv_desc(Nn, H, W, Ci, w.shape[0], w.shape[2], conv.stride[0], conv.padding[0])
    y, st = ops.conv2d_fwd(xg, ops.pack_weight(w.cuda(), d), d, want_stats=True)
    C = y.shape[-1]
    rm, rv, nbt = torch.zeros(C).cuda(), torch.ones(C).cuda(), torch.zeros((), dtype=torch.int64).cuda()
    z, _, _ = ops.bn_train_fwd(y, st, bn.weight.detach().cuda(), bn.bias.detach().cuda(), rm, rv, nbt, res=res, relu=relu)
    return y, z


with torch.no_grad():
    b = ref.backbone
    # oracle (emulated) and GPU side by side
    ze = _bn_train(F.conv2d(q(x), q(b.conv1.weight), stride=2, padding=3), b.bn1)
    _, zg = gpu_conv_bn(ops.to_nhwc_bf16(x.cuda(), 8), b.conv1, b.bn1)
    print("stem", rel(nc(zg), ze))
    ze = F.max_pool2d(ze, 3, 2, 1)
    zg, _ = ops.maxpool3x3s2_fwd(zg)
    print("pool", rel(nc(zg), ze))
    for li, layer in enumerate((b.layer1, b.layer2, b.layer3, b.layer4)):
        for bi, blk in enumerate(layer):
            idt_e, idt_g = ze, zg
            z1e = _bn_train(F.conv2d(ze, q(blk.conv1.weight)), blk.bn1)
            _, z1g = gpu_conv_bn(zg, blk.conv1, blk.bn1)
            z2e = _bn_train(F.conv2d(z1e, q(blk.conv2.weight), stride=blk.conv2.stride, padding=1), blk.bn2)
            _, z2g = gpu_conv_bn(z1g, blk.conv2, blk.bn2)
            if blk.downsample is not None:
                idt_e = _bn_train(F.conv2d(ze, q(blk.downsample[0].weight), stride=blk.downsample[0].stride), blk.downsample[1], relu=False)
                _, idt_g = gpu_conv_bn(zg, blk.downsample[0], blk.downsample[1], relu=False)
            ze = _bn_train(F.conv2d(z2e, q(blk.conv3.weight)), blk.bn3, res=idt_e)
            _, zg = gpu_conv_bn(z2g, blk.conv3, blk.bn3, res=idt_g)
            print(f"layer{li+1}.{bi}: z1 {rel(nc(z1g), z1e):.4f} z2 {rel(nc(z2g), z2e):.4f} idt {rel(nc(idt_g), idt_e):.4f} out {rel(nc(zg), ze):.4f}")
    up = ref.upsampling
    for i in (0, 3, 6):
        ze = _bn_train(F.conv_transpose2d(ze, q(up[i].weight), stride=2, padding=1), up[i + 1])
        _, zg = gpu_conv_bn(zg, up[i], up[i + 1], transposed=True)
        print(f"up{i}", rel(nc(zg), ze))
    he = F.conv2d(ze, q(ref.head.weight), ref.head.bias)
    d = ops.conv_desc(N, zg.shape[1], zg.shape[2], 256, 16, 1)
    hg = ops.conv2d_fwd(zg, ops.pack_weight(ref.head.weight.detach().cuda(), d), d, bias=ref.head.bias.detach().cuda(), out_f32=True)
    print("head (chained kernels vs emulated)", rel(nc(hg), he))
    net = pr._pose_resnet("t", 16, pr.Bottleneck_default, layers, False, False)
    net.load_state_dict(ref.state_dict())
    net = net.cuda().train()
    yx = net(x.cuda()).cpu()
    print("executor vs chained kernels", rel(yx, nc(hg)), " executor vs emulated", rel(yx, he), " emulated vs fp32", rel(he, ref(x)))
