# Dev tool: SQ counters of the style network's convolution kernels (one pass of tools/time_style_layers.py under rocprofv3 --pmc, program directly after --)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/ps1 /tmp/ps2
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/ps1 -- python3 tools/time_style_layers.py 32 > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d /tmp/ps2 -- python3 tools/time_style_layers.py 32 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
from collections import defaultdict
for d in ("/tmp/ps1", "/tmp/ps2"):
    fs = glob.glob(d + "/*/*counter_collection.csv")
    if not fs:
        print(d, "no counters"); continue
    acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(int)
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        if "patch3x3" not in k and "igemm" not in k: continue
        key = (k[:70], r["Grid_Size"])
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
    for key, v in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", kv[1].get("SQ_INSTS_MFMA", 0)))[:8]:
        print(key)
        for c, x in v.items(): print(f"    {c:28s} {x:18.0f}")
PY
