"""Dev tool: aggregate a rocprofv3 kernel trace by (kernel, grid) -> ms/step."""
import csv, glob, collections, sys
d = sys.argv[1]; steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
f = (glob.glob(d + '/*/*_kernel_trace.csv') + glob.glob(d + '/*_kernel_trace.csv'))[0]
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(lambda: [0, 0])
tot = collections.defaultdict(lambda: [0, 0])
for r in rows:
    n = r['Kernel_Name']
    dur = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    short = n.split('::')[-1][:34] if '::' in n else n[:34]
    tot[short][0] += 1; tot[short][1] += dur
    if 'igemm' in n or 'wgrad' in n:
        key = (short, int(r['Grid_Size_X']) // 256, r['Grid_Size_Y'], r['Grid_Size_Z'])
        agg[key][0] += 1; agg[key][1] += dur
print("== totals per kernel (ms/step)")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1])[:16]:
    print(f"{v[1]/steps/1e6:8.3f}  n/step={v[0]/steps:6.1f} avg={v[1]/v[0]/1e3:7.1f}us  {k}")
print("== MFMA kernels by grid")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
    print(f"{v[1]/steps/1e6:8.3f}  n/step={v[0]/steps:6.1f} avg={v[1]/v[0]/1e3:7.1f}us  {k}")
