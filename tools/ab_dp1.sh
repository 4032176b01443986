#!/bin/bash
# Dev tool: the data-parallel step at ONE rank over a real RCCL process group (UDAPOSE_FORCE_DIST=1), per form, against the plain one-rank step.
P='import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        d = json.loads(l); print(d["ms_per_step"], d["value"], "rccl", d["rccl_ranks"], "|", d["launch"][:100])'
for a in "" "--capture-comm" "--capture-comm --one-bucket" "--grad-comm bf16" "--capture-comm --one-bucket --grad-comm bf16"; do
  echo "== FORCE_DIST $a"
  UDAPOSE_FORCE_DIST=1 timeout -k 10 200 python bench.py --steps 40 --spinup 3 --no-cpu-baseline --no-other-configs $a 2> gpurun_out/r5_cc.err | python -c "$P"
  grep -i "warn\|error\|Traceback" gpurun_out/r5_cc.err | head -5
done
echo "== plain"
timeout -k 10 200 python bench.py --steps 40 --spinup 3 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "$P"
