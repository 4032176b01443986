"""Dev tool: what the memory system gives plain streaming kernels (fill = write only, copy = read + write, sum = read only), 268 MB tensors."""
import torch
n = 268 * 1024 * 1024 // 2
x = torch.empty(n, dtype=torch.bfloat16, device="cuda")
y = torch.empty_like(x)
def t(fn, k=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(k): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / k * 1e3
mb = n * 2 / 1e6
us = t(lambda: x.fill_(1.0)); print(f"fill  {us:7.1f} us  {mb / us:6.2f} TB/s written")
us = t(lambda: y.copy_(x)); print(f"copy  {us:7.1f} us  {2 * mb / us:6.2f} TB/s read+written")
us = t(lambda: x.view(torch.int16).sum()); print(f"sum   {us:7.1f} us  {mb / us:6.2f} TB/s read")
