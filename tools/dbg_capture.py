"""Dev tool: which cross-stream event patterns survive hipStreamEndCapture on this ROCm (pure torch)."""
import faulthandler, sys
faulthandler.enable()
import torch
case = sys.argv[1]
x = torch.ones(1 << 16, device='cuda')
a, b = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
def op(s, t):
    with torch.cuda.stream(s):
        t.mul_(1.0001)
ya, yb = x.clone(), x.clone()
ev1, ev2 = torch.cuda.Event(), torch.cuda.Event()
with torch.cuda.graph(g):
    o = torch.cuda.current_stream()
    a.wait_stream(o); b.wait_stream(o)
    if case == "T1":
        op(a, ya); b.wait_stream(a); op(b, yb)
    elif case == "T2":
        for _ in range(4):
            op(a, ya); b.wait_stream(a); op(b, yb); a.wait_stream(b)
    elif case == "T3":
        for _ in range(4):
            op(a, ya); ev1.record(a); b.wait_event(ev1); op(b, yb); ev2.record(b); a.wait_event(ev2)
    elif case == "T4":
        for _ in range(4):
            x.mul_(1.0001); ev1.record(o); a.wait_event(ev1); op(a, ya); ev2.record(a); o.wait_event(ev2)
    elif case == "T5":   # sibling fork with reused event, join later through a different event (the wgrad pattern)
        evs = [torch.cuda.Event() for _ in range(3)]
        for i in range(6):
            op(a, ya); ev1.record(a); b.wait_event(ev1); op(b, yb); evs[i % 3].record(b)
            if i >= 2: a.wait_event(evs[(i - 2) % 3])
        ev2.record(b); a.wait_event(ev2)
    elif case == "T6":   # same as T5 with fresh events
        pend = []
        for i in range(6):
            op(a, ya); e = torch.cuda.Event(); e.record(a); b.wait_event(e); op(b, yb); e2 = torch.cuda.Event(); e2.record(b); pend.append(e2)
            if i >= 2: a.wait_event(pend[i - 2])
        e = torch.cuda.Event(); e.record(b); a.wait_event(e)
    elif case == "T7":   # one-directional sibling forks, fresh events; both joined into the origin only
        for _ in range(6):
            op(a, ya); b.wait_stream(a); op(b, yb)
    elif case == "T8":   # one-directional, one reused event
        for _ in range(6):
            op(a, ya); ev1.record(a); b.wait_event(ev1); op(b, yb)
    o.wait_stream(a); o.wait_stream(b)
print(case, "capture ok", flush=True)
g.replay(); torch.cuda.synchronize()
print(case, "replay ok", float(ya[0]), float(yb[0]), flush=True)
