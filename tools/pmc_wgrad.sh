# Dev tool: SQ / TCC counters of one weight-gradient layer.  usage: bash tools/pmc_wgrad.sh "<one_wgrad.py args>" (program directly after --)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ARGS=${1:-"2048 8 512 512 3 16 0"}
rm -rf /tmp/pw1 /tmp/pw2 /tmp/pw3 /tmp/pw4
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pw1 -- python3 tools/one_wgrad.py $ARGS > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d /tmp/pw2 -- python3 tools/one_wgrad.py $ARGS > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pw3 -- python3 tools/one_wgrad.py $ARGS > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pw4 -- python3 tools/one_wgrad.py $ARGS > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
from collections import defaultdict
for d in ("/tmp/pw1", "/tmp/pw2", "/tmp/pw3", "/tmp/pw4"):
    fs = glob.glob(d + "/*/*counter_collection.csv")
    if not fs:
        print(d, "no counters"); continue
    acc = defaultdict(list)
    dur = 0
    for r in csv.DictReader(open(fs[0])):
        if "wgrad" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    for k, v in acc.items():
        print(f"{k:28s} {v[-1]:16.0f}")
    print("last launch us", dur)
PY
