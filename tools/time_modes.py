"""Forward time of PoseResNet-101 (N=32, 256x256) and of one Style_net.Net pass per precision mode (HIP events, one stream)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import uda_poseestimation_amd.lib.models as models
from uda_poseestimation_amd.lib.models import Style_net


def timeit(fn, n=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    modes = sys.argv[1].split(",") if len(sys.argv) > 1 else ["bf16", "fp16", "fp32", "f16x2"]
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = models.pose_resnet101(16, pretrained_backbone=False).to(dev)
    x = torch.randn(N, 3, 256, 256, device=dev)
    out = {}
    for train in (True, False):
        net.train(train)
        for m in modes:
            net.precision = m
            with torch.no_grad():
                t = timeit(lambda: net(x))
            out[f"pose101 N={N} {'train' if train else 'eval'} {m}"] = round(t, 3)
            print(f"pose101 N={N} {'train' if train else 'eval'} {m}: {t:.3f} ms  ({N * 24.165e9 / t / 1e9:.1f} TFLOP/s)", flush=True)
    sn = Style_net.Net(Style_net.vgg.to(dev), Style_net.decoder.to(dev)).to(dev)
    c, s = torch.rand(N, 3, 256, 256, device=dev), torch.rand(N, 3, 256, 256, device=dev)
    for m in modes:
        if m == "fp16":
            continue
        sn.precision = m
        try:
            t = timeit(lambda: sn(c, s, 0.5), n=3, warm=1)
        except Exception as e:
            print("style", m, "failed:", e)
            continue
        out[f"style N={N} {m}"] = round(t, 3)
        print(f"style N={N} {m}: {t:.3f} ms  ({N * 94.9e9 / t / 1e9:.1f} TFLOP/s minimal-count)", flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
