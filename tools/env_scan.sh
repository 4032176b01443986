#!/bin/bash
# Dev tool: the headline bench under HIP runtime environment knobs (one per run).  usage: tools/env_scan.sh "VAR=VAL" "VAR=VAL" ...
P='import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        d = json.loads(l); print(d["ms_per_step"], d["value"], "synced", d["ms_per_step_synced"])'
for kv in "$@"; do
  echo -n "[$kv] "
  env $kv timeout -k 10 200 python bench.py --steps 40 --warmup 3 --spinup 3 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "$P" || echo failed
done
