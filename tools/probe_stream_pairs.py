"""Dev probe (round 5): which pairs of torch streams run kernels CONCURRENTLY on this HIP stack (streams map onto GPU_MAX_HW_QUEUES hardware queues,
and hardware queues onto the compute pipes: two streams that share a queue - or a pipe - serialise)?  Pairwise: a spin kernel of T on each of the two
streams, started together; wall ~T = concurrent (.), ~2T = serialised (X)."""
import os, sys, time
import torch
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
streams = [torch.cuda.current_stream()] + [torch.cuda.Stream(device=dev) for _ in range(n)]
CY = 2_000_000
def t_one(s):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        a.record(); torch.cuda._sleep(CY); b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b)
T = min(t_one(streams[1]) for _ in range(3))
print(f"GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')}: one spin kernel {T:.3f} ms; rows / columns = default stream, then {n} pool streams in creation order")
def pair(i, j):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(streams[i]):
        torch.cuda._sleep(CY)
    with torch.cuda.stream(streams[j]):
        torch.cuda._sleep(CY)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3
for i in range(len(streams)):
    row = ""
    for j in range(len(streams)):
        if i == j:
            row += " -"
        else:
            w = min(pair(i, j) for _ in range(2))
            row += " ." if w < 1.5 * T else " X"
    print(f"{i:2d} {row}")
