"""Dev tool: per (kernel, grid) HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KiB per dispatch).
usage: pmc_by_shape.py <fetch_dir> <write_dir> <out.txt> [steps]"""
import csv, glob, re, sys
from collections import defaultdict


def load(d, counter):
    f = (glob.glob(d + '/*/*counter_collection.csv') + glob.glob(d + '/*counter_collection.csv'))[0]
    out = defaultdict(lambda: [0, 0.0, 0.0])
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != counter:
            continue
        n = re.sub(r"^void |\(anonymous namespace\)::|_ZN12_GLOBAL__N_1", "", r['Kernel_Name'])
        n = n.split("(")[0][:64]
        k = (n, int(r['Grid_Size']))
        e = out[k]
        e[0] += 1
        e[1] += float(r['Counter_Value'])
        e[2] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    return out


fe, wr = load(sys.argv[1], 'FETCH_SIZE'), load(sys.argv[2], 'WRITE_SIZE')
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 6
rows = []
for k, (n, kib, us) in fe.items():
    w = wr.get(k, [0, 0.0, 0.0])
    f_mb, w_mb = 2 * kib * 1024 / n / 1e6, (w[1] * 1024 / max(w[0], 1)) / 1e6
    rows.append((n * (f_mb + w_mb), k, n, f_mb, w_mb, us / n))
rows.sort(reverse=True)
with open(sys.argv[3], "w") as o:
    o.write(f"# per (kernel, grid) HBM traffic of the launches of {steps} eager steps (bench.py --eager, separate rocprofv3 --pmc FETCH_SIZE and\n"
            "# --pmc WRITE_SIZE passes; FETCH_SIZE doubled per the gfx950 note of the MI355X guide), sorted by total bytes.\n"
            "# kernel | grid threads | launches | fetch MB (x2) per launch | write MB per launch | avg us (under the profiler) | TB/s\n")
    tot = 0.0
    for t, (name, grid), n, f_mb, w_mb, us in rows:
        tot += t
        o.write(f"{name:64s} | {grid:9d} | {n:5d} | {f_mb:8.1f} | {w_mb:8.1f} | {us:8.1f} | {(f_mb + w_mb) / max(us, 1e-9):5.2f}\n")
    o.write(f"# total {tot / 1e3 / steps:.1f} GB per step over all kernels\n")
print("total GB/step", tot / 1e3 / steps)
