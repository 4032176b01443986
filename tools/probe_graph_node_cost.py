"""Dev probe (round 5): what ONE kernel node costs inside a replayed hipGraph on this stack - a chain of dependent tiny kernels on one stream,
and three such chains forked from / joined into the origin stream (the step's structure: ~1270 nodes in three chains)."""
import sys, time
import torch
dev = torch.device("cuda:0")
x = [torch.zeros(256, device=dev) for _ in range(3)]
big = [torch.zeros(1 << 22, device=dev) for _ in range(3)]


def chain(t, n):
    for _ in range(n):
        t.add_(1.0)


def build(nchains, n, tensors):
    g = torch.cuda.CUDAGraph()
    side = [torch.cuda.Stream() for _ in range(nchains - 1)]
    with torch.cuda.graph(g):
        main = torch.cuda.current_stream()
        for s in side:
            s.wait_stream(main)
        chain(tensors[0], n)
        for i, s in enumerate(side):
            with torch.cuda.stream(s):
                chain(tensors[i + 1], n)
        for s in side:
            main.wait_stream(s)
    return g


def timeit(g, reps=20):
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


for label, tensors in (("256-element add (latency only)", x), ("16 MiB add (a 10 us streaming kernel)", big)):
    for nch in (1, 2, 3):
        for n in (100, 400):
            us = timeit(build(nch, n, tensors))
            print(f"{label}: {nch} chain(s) x {n} nodes: {us:9.1f} us per replay = {us / n:6.2f} us per node of a chain", flush=True)

# --- the same chains as SEPARATE single-chain graphs, one per stream, launched together and joined by events per iteration
def separate(nch, n, tensors, reps=20):
    streams = [torch.cuda.Stream() for _ in range(nch)]
    graphs = []
    for i in range(nch):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            chain(tensors[i], n)
        graphs.append(g)
    evs = [torch.cuda.Event() for _ in range(nch)]
    main = torch.cuda.current_stream()
    def one():
        e0 = torch.cuda.Event(); e0.record(main)
        for s, g, e in zip(streams, graphs, evs):
            s.wait_event(e0)
            with torch.cuda.stream(s):
                g.replay()
            e.record(s)
        for e in evs:
            main.wait_event(e)
    for _ in range(3):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        one()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


def eager(nch, n, tensors, reps=5):
    streams = [torch.cuda.Stream() for _ in range(nch)]
    main = torch.cuda.current_stream()
    def one():
        for i, s in enumerate(streams):
            s.wait_stream(main)
            with torch.cuda.stream(s):
                chain(tensors[i], n)
        for s in streams:
            main.wait_stream(s)
    one(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        one()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


for label, tensors in (("256-element add", x), ("16 MiB add", big)):
    for nch in (1, 2, 3):
        us = separate(nch, 400, tensors)
        print(f"{label}: {nch} SEPARATE single-chain graph(s) x 400 nodes on {nch} stream(s): {us:9.1f} us per iteration = {us / 400:6.2f} us per node of a chain", flush=True)
    us = eager(3, 400, tensors)
    print(f"{label}: 3 streams x 400 EAGER launches: {us:9.1f} us per iteration = {us / 400:6.2f} us per node of a chain", flush=True)
