"""Probe (round 5): a hipMemsetAsync node followed by the kernel node that depends on it, captured into a small hipGraph and replayed with
eager device work between the replays.  On this image (ROCm 7.2.0, torch 2.10) the THIRD and later replays run the clear AFTER the kernel
(the buffer reads 0 instead of 0 + 1) with the runtime's default DEBUG_CLR_GRAPH_PACKET_CAPTURE=1; with that variable 0, or with the clear
done by a kernel (`zero`), every replay is right.  Nothing of the package is involved: torch + libamdhip64 only.  This is why every clear
on a capturable path of csrc/ goes through pw_zero (common.h) instead of hipMemsetAsync.

`copy`: the same question for a memcpy node (dx.copy_(zeros): torch issues hipMemcpyAsync for a contiguous device-to-device copy).

usage: python tools/probe/graph_memset_order.py [zero|copy]        (prints |d - 1| per replay: 0 = right, 1 = the clear came last)
"""
import ctypes as C
import sys
import torch

dev = torch.device("cuda:0")
hip = C.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
n = 65536


def clear_then_add(g):
    dx = torch.empty_like(g)
    if "zero" in sys.argv:
        dx.zero_()              # (a fill kernel)
    elif "copy" in sys.argv:
        dx.copy_(zeros)         # (a memcpy node)
    else:
        rc = hip.hipMemsetAsync(dx.data_ptr(), 0, dx.numel() * 4, torch.cuda.current_stream().cuda_stream)
        assert rc == 0
    dx += g                     # (stands for the scatter kernel that accumulates into the cleared buffer)
    return dx


zeros = torch.zeros(n, device=dev)
for trial in range(3):
    stream = torch.cuda.Stream()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=stream, capture_error_mode="global"):
        d = clear_then_add(torch.ones(n, device=dev))
    res = []
    for rep in range(6):
        with torch.cuda.stream(stream):
            gr.replay()
        torch.cuda.synchronize()
        tmp = (d - 1).abs()
        res.append("%.3g" % tmp.max().item())
        del tmp
    print("graph", trial, "|d - 1| per replay:", res, flush=True)
