"""Dev tool: per-layer time of one style-network pass (encoder on content, decoder) per precision: where the style pass spends its time."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uda_poseestimation_amd import ops
from uda_poseestimation_amd.lib.models import Style_net
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
from uda_poseestimation_amd import _hip
TILE = int(sys.argv[2]) if len(sys.argv) > 2 else -1
PATCH = int(sys.argv[3]) if len(sys.argv) > 3 else None          # policy patch_conv override of the style network's own policy
POL = _hip.policy(igemm_tile=TILE) if TILE >= 0 else None        # None: the style network's own policy
dev = torch.device("cuda:0")
Style_net.vgg.to(dev); Style_net.decoder.to(dev)
if PATCH is not None:
    Style_net._SeqRunner.policy_overrides = dict(Style_net._SeqRunner.policy_overrides, patch_conv=PATCH)
net = Style_net.Net(torch.nn.Sequential(*list(Style_net.vgg.children())[:31]), Style_net.decoder).to(dev)
img = torch.rand(N, 3, 256, 256, device=dev)
def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for prec in (("bf16",) if TILE >= 0 else ("bf16", "f16x2")):
    net.precision = prec
    tot = 0.0
    for name, runner, x in (("enc", net._enc, net._image_in(img)), ("dec", net._dec, None)):
        if x is None:
            x = feat
        for si, st in enumerate(runner.steps):
            f32 = 'split' if x.dtype == ops.SPLIT else x.dtype == torch.float32
            if st.kind == "pool":
                us = timeit(lambda: ops.maxpool2x2_ceil(x)); y = ops.maxpool2x2_ceil(x); desc = "pool"
                fl = 0
            else:
                Nn, H, W, Cin = x.shape
                d = ops.conv_desc(Nn, H, W, Cin, st.conv.out_channels, 3, 1, 1, reflect=True, upsample=st.upsample, policy=POL if POL is not None else runner.policy())
                w, b = runner._packed(st, d, f32)
                last = name == "dec" and si == len(runner.steps) - 1
                us = timeit(lambda: ops.conv2d_fwd(x, w, d, bias=b, relu=st.relu, out_f32=last))
                y = ops.conv2d_fwd(x, w, d, bias=b, relu=st.relu, out_f32=last)
                Ho = H * (2 if st.upsample else 1)
                fl = 2.0 * Nn * Ho * Ho * st.conv.out_channels * (3 if Cin == 8 else Cin) * 9
                desc = f"conv {Cin}->{st.conv.out_channels} @{Ho}{' up' if st.upsample else ''}"
            tot += us
            print(f"{prec} {name}{si:2d} {desc:28s} {us:8.1f} us  {fl / us / 1e6 if fl else 0:7.0f} TFLOP/s", flush=True)
            x = y
        if name == "enc":
            feat = x
    print(f"{prec} total (one encoder pass + decoder) {tot / 1e3:.2f} ms\n", flush=True)
