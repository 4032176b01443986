"""Dev tool: sum FETCH_SIZE (KiB at the L2's fabric side) and launch count per kernel name prefix from a rocprofv3 --pmc FETCH_SIZE CSV."""
import csv, glob, sys, re
f = sorted(glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True))[0]
agg = {}
for r in csv.DictReader(open(f)):
    if r['Counter_Name'] != sys.argv[2]:
        continue
    n = r['Kernel_Name']
    key = 'bn_apply_chunk' if 'bn_apply_chunk' in n else ('igemm lean 1x1 fwd' if ('igemm_kernel' in n and 'Lb0ELb0ELi3' in n) else ('igemm 3x3 run-staged' if ('igemm_kernel' in n and 'Lb0ELb0ELi1' in n) else ('bn_bwd_apply_pre_chunk' if 'bn_bwd_apply_pre_chunk' in n else None)))
    if key is None:
        continue
    c, s = agg.get(key, (0, 0.0))
    agg[key] = (c + 1, s + float(r['Counter_Value']))
for k, (c, s) in sorted(agg.items()):
    print(f"{k:28s} launches {c:6d}  {sys.argv[2]} per launch {s / c:10.1f}")
