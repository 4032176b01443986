"""Dev tool: SQ counters per (kernel, grid) from one rocprofv3 --pmc pass of an eager bench run: MFMA-pipe busy fraction and the split of
wave cycles into issuing / issue-stalled / parked, for the most expensive igemm / weight-gradient shapes.
usage: pmc_sq_by_shape.py <pmc_dir> <out.txt> [top]
MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x kernel cycles at the measured GRBM clock is not collected here: the
duration x 2.0 GHz nominal under load is used and stated); wave fractions = SQ_ACTIVE_INST_ANY | SQ_WAIT_INST_ANY | SQ_WAIT_ANY over
SQ_WAVE_CYCLES (quad-cycle units, disjoint: MI355X guide, rocprofv3 PMC slots)."""
import csv, glob, re, sys
from collections import defaultdict
d, out = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 14
f = (glob.glob(d + '/*/*counter_collection.csv') + glob.glob(d + '/*counter_collection.csv'))[0]
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(int)
dur = defaultdict(float)
seen = set()
for r in csv.DictReader(open(f)):
    n = re.sub(r"^void |\(anonymous namespace\)::|_ZN12_GLOBAL__N_1", "", r['Kernel_Name']).split("(")[0][:64]
    if 'igemm' not in n and 'wgrad' not in n:
        continue
    k = (n, int(r['Grid_Size']), int(r.get('LDS_Block_Size', 0) or 0))
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    did = r['Dispatch_Id']
    if (k, did) not in seen:
        seen.add((k, did))
        cnt[k] += 1
        dur[k] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
rows = sorted(acc.items(), key=lambda kv: -dur[kv[0]])[:top]
with open(out, "w") as o:
    o.write("# SQ counters of the most expensive igemm / weight-gradient launch shapes (one rocprofv3 --pmc pass over eager steps of bench.py; durations\n"
            "# are under the profiler).  MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x duration x 2.0 GHz); wave-cycle split over SQ_WAVE_CYCLES.\n"
            "# kernel | grid threads | LDS B | launches | avg us | MFMA busy | issuing | issue-stalled | parked (s_waitcnt / barrier) | VALU per MFMA\n")
    for k, c in rows:
        n = cnt[k]
        us = dur[k] / n
        wc = c.get('SQ_WAVE_CYCLES', 0.0)
        busy = c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / n / (1024 * us * 2000.0) if us else 0.0
        fr = lambda name: (c.get(name, 0.0) / wc) if wc else 0.0
        vpm = c.get('SQ_INSTS_VALU', 0.0) / max(c.get('SQ_INSTS_MFMA', 0.0), 1.0)
        o.write(f"{k[0]:64s} | {k[1]:8d} | {k[2]:6d} | {n:4d} | {us:7.1f} | {busy:5.1%} | {fr('SQ_ACTIVE_INST_ANY'):5.1%} | {fr('SQ_WAIT_INST_ANY'):5.1%} | "
                f"{fr('SQ_WAIT_ANY'):5.1%} | {vpm:5.2f}\n")
print(open(out).read())
