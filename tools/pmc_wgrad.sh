cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pw1 -- python3 tools/one_wgrad.py 2048 8 512 16 > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d /tmp/pw2 -- python3 tools/one_wgrad.py 2048 8 512 16 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
from collections import defaultdict
for d in ("/tmp/pw1", "/tmp/pw2"):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    acc = defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "wgrad" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    for k, v in acc.items():
        print(f"{k:28s} {v[-1]:16.0f}")
    print("last launch us", dur)
PY
