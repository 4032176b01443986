"""Dev tool: reconstruct ONE captured step's device timeline from a `rocprofv3 --kernel-trace` CSV (kernel start / end stamps survive
hipGraph replay): where the gradient chains end, where the weight-gradient launches run, what is exposed at the step's end.

usage: python tools/step_timeline.py <dir or *_kernel_trace.csv> [step index from the end, default 30]
Steps are delimited by the optimizer tail (opt_tail_k: one launch per step)."""
import csv
import glob
import os
import sys


def classify(name):
    if "wgrad_persist" in name:
        return "wg_persist"
    if "wgrad" in name:
        return "wg_group"
    if "opt_tail" in name:
        return "tail"
    if "igemm" in name:
        return "igemm"
    if "bn_bwd" in name:
        return "bn_bwd"
    if "bn_" in name:
        return "bn_fwd"
    return "other"


def main():
    src = sys.argv[1]
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    if os.path.isdir(src):
        src = sorted(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True))[0]
    rows = []
    with open(src) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", ""), r.get("Stream_Id", "")))
    rows.sort()
    tails = [i for i, r in enumerate(rows) if "opt_tail" in r[2]]
    if len(tails) < back + 2:
        back = len(tails) - 2
    a, b = tails[-back - 2], tails[-back - 1]
    step = rows[a + 1:b + 1]
    # (kernels of the step: everything that starts after the previous tail's start and up to this tail)
    t0 = rows[a][1]
    print(f"step of {len(step)} kernels, {(rows[b][1] - t0) / 1e3:.1f} us from the previous tail's end to this tail's end")
    cats = {}
    for s, e, n, q, st in step:
        c = classify(n)
        d = cats.setdefault(c, [s, e, 0, 0])
        d[0] = min(d[0], s); d[1] = max(d[1], e); d[2] += 1; d[3] += e - s
    for c, (s, e, k, tot) in sorted(cats.items(), key=lambda kv: kv[1][0]):
        print(f"  {c:11s} n={k:5d} first start {(s - t0) / 1e3:9.1f} us  last end {(e - t0) / 1e3:9.1f} us  kernel time {tot / 1e3:9.1f} us")
    print("  weight-gradient launches (start, end, us, grid if known):")
    for s, e, n, q, st in step:
        if "wgrad" in n:
            print(f"    {(s - t0) / 1e3:9.1f} -> {(e - t0) / 1e3:9.1f}  {(e - s) / 1e3:8.1f} us  q{q} {n[20:70]}")
    # busy union of the backward chain kernels after the first bn_bwd launch
    bb = [r for r in step if classify(r[2]) == "bn_bwd"]
    if bb:
        print(f"  backward chains: first bn_bwd start {(bb[0][0] - t0) / 1e3:.1f} us, last bn_bwd end {(max(r[1] for r in bb) - t0) / 1e3:.1f} us")


if __name__ == "__main__":
    main()
