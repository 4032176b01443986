"""Dev tool: read a rocprofv3 kernel trace (csv) of a graph-replayed bench run and print, for the last full steps, how busy the
device was: time with 0 / 1 / 2 / 3+ kernels in flight, phase boundaries (forward | losses | backward | tail) and the largest gaps.
usage: python tools/timeline.py <kernel_trace.csv> [n_steps]"""
import csv, sys
from collections import defaultdict

path = sys.argv[1]
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = []
with open(path) as f:
    r = csv.DictReader(f)
    for d in r:
        rows.append((int(d["Start_Timestamp"]), int(d["End_Timestamp"]), d["Kernel_Name"], d.get("Queue_Id", "0"), d.get("Stream_Id", "0")))
rows.sort()
tails = [i for i, x in enumerate(rows) if "opt_tail_k" in x[2]]
print("kernels", len(rows), "tail launches", len(tails))
for si in range(len(tails) - nsteps, len(tails)):
    lo, hi = tails[si - 1] + 1, tails[si]
    step = rows[lo:hi + 1]
    t0, t1 = step[0][0], max(x[1] for x in step)
    ev = []
    for s, e, *_ in step:
        ev.append((s, 1)); ev.append((e, -1))
    ev.sort()
    busy = defaultdict(int)
    depth, last = 0, t0
    gaps = []
    for t, dlt in ev:
        busy[min(depth, 4)] += t - last
        if depth == 0 and t - last > 3000:
            gaps.append((t - last, last - t0))
        last = t
        depth += dlt
    tot = t1 - t0
    first_bwd = next((x for x in step if "ELb1E" in x[2] or "bn_bwd" in x[2]), None)
    last_fwd_conv = None
    for x in step:
        if first_bwd and x[0] >= first_bwd[0]:
            break
        if "igemm" in x[2]:
            last_fwd_conv = x
    wg = [x for x in step if "wgrad" in x[2]]
    print(f"step {si}: {tot/1e6:.3f} ms, {len(step)} kernels; in flight 0/1/2/3/4+: " + " ".join(f"{busy[k]/tot*100:.1f}%" for k in range(5)))
    if first_bwd and last_fwd_conv:
        print(f"   last fwd conv ends {(last_fwd_conv[1]-t0)/1e6:.3f} ms; first bwd kernel starts {(first_bwd[0]-t0)/1e6:.3f} ms; "
              f"wgrad launches at " + ", ".join(f"{(x[0]-t0)/1e6:.2f}-{(x[1]-t0)/1e6:.2f}" for x in wg) + f"; tail starts {(step[-1][0]-t0)/1e6:.3f}")
    gaps.sort(reverse=True)
    print("   idle gaps > 3 us: n =", len(gaps), "sum %.3f ms;" % (sum(g[0] for g in gaps) / 1e6), "largest:", [(round(g[0] / 1e3, 1), round(g[1] / 1e6, 2)) for g in gaps[:6]])
    # kernel-time by queue
    byq = defaultdict(int)
    for s, e, n, q, st in step:
        byq[(q, st)] += e - s
    print("   busy ms per (queue, stream):", {k: round(v / 1e6, 2) for k, v in sorted(byq.items())})
