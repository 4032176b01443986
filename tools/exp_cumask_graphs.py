"""Dev tool (round 5): three PoseResNet-101 train-mode forward chains (N = 32, 256x256, bf16: the step's forward phase),
each captured into its OWN hipGraph and replayed on its own stream - ordinary streams (time sharing of every CU, what the one-graph step
does) against CU-masked streams (hipExtStreamCreateWithCUMask; a graph launched on a masked stream runs on that mask:
tools/probe/cumask_graph.hip).  Per iteration the three replays start together and the iteration ends when all three have ended.

usage: python tools/exp_cumask_graphs.py [fwd|fwdbwd] SPEC [SPEC ...]      SPEC = plain | lo-hi,lo-hi,lo-hi  (CUs [lo, hi) of every XCC per stream)
"""
import ctypes as C
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import uda_poseestimation_amd.lib.models as models
from uda_poseestimation_amd.lib.models.pose_resnet import PoseResNet

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
hip = C.CDLL("libamdhip64.so")
PoseResNet.default_precision = "bf16"
mode = sys.argv[1]
specs = sys.argv[2:]


def masked_stream(lo, hi):
    m = (C.c_uint32 * 8)()
    for b in range(256):
        if lo <= b // 8 < hi:
            m[b >> 5] |= 1 << (b & 31)
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), 8, m)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value, device=dev)


torch.manual_seed(0)
nets = [models.pose_resnet101(16, pretrained_backbone=False).to(dev).train() for _ in range(3)]
for net in nets:        # (POL="field=int,field=int": dispatch-policy overrides for all three chains)
    net.policy.update({k_: int(v_) for k_, v_ in (a_.split("=") for a_ in os.environ.get("POL", "").split(",") if a_)})
xs = [torch.randn(32, 3, 256, 256, device=dev) for _ in range(3)]
graphs = []
side = torch.cuda.Stream(device=dev)
for i, (net, x) in enumerate(zip(nets, xs)):
    bwd = mode == "fwdbwd" and i > 0          # (stream 0 = the teacher: forward only; streams 1, 2 = the student passes: forward + gradient chain + weight gradients)
    with torch.cuda.stream(side):
        for _ in range(2):
            if bwd:
                y = net(x); y.backward(torch.ones_like(y) * 1e-3)
            else:
                with torch.no_grad():
                    net(x)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    net._capture_token = object()
    with torch.cuda.graph(g):
        if bwd:
            y = net(x); y.backward(torch.ones_like(y) * 1e-3)
        else:
            with torch.no_grad():
                y = net(x)
    net._capture_token = None
    graphs.append(g)
torch.cuda.synchronize()


def run(streams, reps=30):
    evs = [torch.cuda.Event() for _ in range(3)]
    def one():
        for s, g in zip(streams, graphs):
            with torch.cuda.stream(s):
                g.replay()
        for s, e in zip(streams, evs):
            e.record(s)
        for s in streams:           # join: the next iteration's replays start together
            for e in evs:
                s.wait_event(e)
    for _ in range(5):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        one()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def alone(stream, g, reps=30):
    with torch.cuda.stream(stream):
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            g.replay()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


# the same three chains as THREE BRANCHES OF ONE GRAPH (forked from / joined into the capture's origin stream: the product's form)
def one_graph():
    g = torch.cuda.CUDAGraph()
    sides = [torch.cuda.Stream(device=dev) for _ in range(2)]
    for net in nets:
        net._capture_token = object()
    with torch.cuda.graph(g):
        main = torch.cuda.current_stream()
        for i, (net, x) in enumerate(zip(nets, xs)):
            bwd = mode == "fwdbwd" and i > 0
            st = main if i == 0 else sides[i - 1]
            if st is not main:
                st.wait_stream(main)
            with torch.cuda.stream(st):
                if bwd:
                    y = net(x); y.backward(torch.ones_like(y) * 1e-3)
                else:
                    with torch.no_grad():
                        y = net(x)
        for st in sides:
            main.wait_stream(st)
    for net in nets:
        net._capture_token = None
    return g


def alone_g(g, reps=30):
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


plain = [torch.cuda.Stream(device=dev) for _ in range(3)]
# device spin-up: an idle MI355X needs seconds of load to reach its steady clocks (a cold first measurement reads 20-25 % slow)
t_spin = time.perf_counter()
while time.perf_counter() - t_spin < 6.0:
    run(plain, reps=10)
if "onegraph" in specs:
    specs = [s_ for s_ in specs if s_ != "onegraph"]
    g1 = one_graph()
    for rnd in range(3):
        print(f"{mode}: the three chains as three branches of ONE graph: {alone_g(g1):.3f} ms | as three separate graphs on three streams: {run(plain):.3f} ms", flush=True)
print(f"{mode}: one chain alone on the chip: " + " ".join(f"{alone(plain[0], g):.2f}" for g in graphs) + " ms", flush=True)
for rnd in range(2):
    for spec in specs:
        if spec == "plain":
            st = plain
        else:
            st = []
            for part in spec.split(","):
                if part == "p":
                    st.append(torch.cuda.Stream(device=dev))
                else:
                    lo, hi = part.split("-")
                    st.append(masked_stream(int(lo), int(hi)))
        print(f"{mode} [{spec}] three chains together: {run(st):.3f} ms", flush=True)
