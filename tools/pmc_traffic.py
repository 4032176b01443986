"""Dev tool: aggregate two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KiB per dispatch) into
profiles/r1_pmc_hbm_traffic.txt: {'fetch': {kernel family: (launches, sum KiB)}, 'write': {...}}.
usage: pmc_traffic.py <fetch_dir> <write_dir> <out.txt>"""
import csv, glob, sys
def agg(d, counter):
    f = (glob.glob(d + '/*/*counter_collection.csv') + glob.glob(d + '/*counter_collection.csv'))[0]
    out = {}
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != counter:
            continue
        n = r['Kernel_Name']
        fam = 'igemm' if 'igemm_kernel' in n else ('wgrad' if 'wgrad' in n else ('bn' if 'bn_' in n else None))
        if fam is None:
            continue
        c, s = out.get(fam, (0, 0.0))
        out[fam] = (c + 1, s + float(r['Counter_Value']))
    return out
res = {'fetch': agg(sys.argv[1], 'FETCH_SIZE'), 'write': agg(sys.argv[2], 'WRITE_SIZE')}
open(sys.argv[3], 'w').write(repr(res) + "\n")
print(res)
