import sys, subprocess, os
PAT = ["Q1", "Q2", "Q3", "Q4", "Q5", "Q6"]
if len(sys.argv) == 1:
    for p in PAT:
        r = subprocess.run([sys.executable, __file__, p], capture_output=True, text=True)
        print(p, "rc", r.returncode, r.stdout.strip()[-200:], r.stderr.strip()[-300:].replace("\n", " | ") if r.returncode else "")
    sys.exit(0)
import torch
p = sys.argv[1]
x = torch.zeros(1 << 20, device="cuda")
ys = [torch.zeros(1 << 20, device="cuda") for _ in range(6)]
a, b, c = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
def work(st, y):
    with torch.cuda.stream(st):
        y.add_(x); y.mul_(2.0)
with torch.cuda.graph(g):
    M = torch.cuda.current_stream()
    x.add_(1.0)
    if p == "Q1":
        a.wait_stream(M); work(a, ys[0]); b.wait_stream(a); work(b, ys[1]); a.wait_stream(b); M.wait_stream(a)
    elif p == "Q2":
        a.wait_stream(M); work(a, ys[0]); b.wait_stream(a); work(b, ys[1]); M.wait_stream(b); M.wait_stream(a)
    elif p == "Q3":
        a.wait_stream(M); b.wait_stream(M); work(a, ys[0]); work(b, ys[1]); a.wait_stream(b); M.wait_stream(a)
    elif p == "Q4":
        a.wait_stream(M); work(a, ys[0]); b.wait_stream(M); b.wait_stream(a); work(b, ys[1]); M.wait_stream(a); M.wait_stream(b)
    elif p == "Q5":
        a.wait_stream(M); work(a, ys[0]); b.wait_stream(M); b.wait_stream(a); work(b, ys[1]); work(a, ys[2]); a.wait_stream(b); M.wait_stream(a)
    elif p == "Q6":
        a.wait_stream(M); b.wait_stream(M); work(a, ys[0]); b.wait_stream(a); work(b, ys[1]); work(a, ys[2]); M.wait_stream(b); M.wait_stream(a)
    elif p == "P1":
        a.wait_stream(M); work(a, ys[0]); b.wait_stream(a); work(b, ys[1]); work(a, ys[2]); a.wait_stream(b); M.wait_stream(a)
    elif p == "P2":
        a.wait_stream(M); work(a, ys[0]); M.wait_stream(a); x.add_(1.0); c.wait_stream(M); work(c, ys[1]); x.add_(1.0); M.wait_stream(c)
    elif p == "P3":
        a.wait_stream(M); work(a, ys[0]); M.wait_stream(a); x.add_(1.0)
        a.wait_stream(M); work(a, ys[0]); b.wait_stream(a); work(b, ys[1]); work(a, ys[2]); a.wait_stream(b); M.wait_stream(a)
    elif p in ("P4", "P5"):
        a.wait_stream(M); work(a, ys[0]); work(M, ys[3]); M.wait_stream(a); x.add_(1.0)
        # "backward": a waits M, chains on a and M
        a.wait_stream(M); work(a, ys[0]); work(M, ys[3]); M.wait_stream(a)
        a.wait_stream(M)
        b.wait_stream(a); work(b, ys[1]); work(a, ys[0])
        c.wait_stream(M); work(c, ys[2]); work(M, ys[3])
        if p == "P5":
            with torch.cuda.stream(c):
                ys[4].zero_()
        a.wait_stream(b); M.wait_stream(c); M.wait_stream(a)
    x.add_(1.0)
g.replay(); torch.cuda.synchronize()
print("ok", float(x[0]))
