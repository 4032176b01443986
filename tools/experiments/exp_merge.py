"""Dev experiment: one fwd+bwd pass of the student at N=64 on one stream vs two concurrent N=32 passes on two streams
(with / without a concurrent teacher forward at N=32), all captured in hipGraphs."""
import sys, time
import torch
sys.path.insert(0, ".")
import uda_poseestimation_amd.lib.models as models

dev = torch.device("cuda:0")
stu = models.pose_resnet101(num_keypoints=16, pretrained_backbone=False).to(dev).train()
tea = models.pose_resnet101(num_keypoints=16, pretrained_backbone=False).to(dev).train()
x64 = torch.randn(64, 3, 256, 256, device=dev)
xa, xb, xt = x64[:32].contiguous(), x64[32:].contiguous(), torch.randn(32, 3, 256, 256, device=dev)
d64 = torch.randn(64, 16, 64, 64, device=dev) * 1e-3
da, db = d64[:32].contiguous(), d64[32:].contiguous()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def merged(with_teacher):
    main = torch.cuda.current_stream()
    stu.zero_grad()
    stu.prepare(x64)
    if with_teacher:
        with torch.no_grad():
            tea.prepare(xt)
        s1.wait_stream(main)
        with torch.cuda.stream(s1), torch.no_grad():
            tea(xt)
    y = stu(x64)
    y.backward(d64)
    if with_teacher:
        main.wait_stream(s1)


def split(with_teacher):
    main = torch.cuda.current_stream()
    stu.zero_grad()
    stu.prepare(xa)
    if with_teacher:
        with torch.no_grad():
            tea.prepare(xt)
        s1.wait_stream(main)
        with torch.cuda.stream(s1), torch.no_grad():
            tea(xt)
    s2.wait_stream(main)
    with torch.cuda.stream(s2):
        yb = stu.forward_deferred_bn(xb)
    ya = stu(xa)
    with torch.cuda.stream(s2):
        yb.backward(db)
    ya.backward(da)
    main.wait_stream(s2)
    stu.apply_deferred_bn()
    stu.finish_grads()
    if with_teacher:
        main.wait_stream(s1)


def timeit(fn, arg, name):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            fn(arg)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    tok = object()
    stu._capture_token = tea._capture_token = tok
    with torch.cuda.graph(g):
        fn(arg)
    stu._capture_token = tea._capture_token = None
    for _ in range(30):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(60):
        g.replay()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 60 * 1e3:.3f} ms", flush=True)


for rep in range(2):
    timeit(merged, False, "student fwd+bwd N=64, one stream            ")
    timeit(split, False, "student fwd+bwd 2 x N=32, two streams        ")
    timeit(merged, True, "  + teacher fwd N=32 concurrently (merged)   ")
    timeit(split, True, "  + teacher fwd N=32 concurrently (2 x N=32) ")
