#!/bin/bash
# Dev tool: A/B several bench configurations on the same box, interleaved.  usage: tools/abn.sh rounds steps "<args 1>" "<args 2>" ...
R="$1"; S="$2"; shift 2
for i in $(seq 1 $R); do
  for spec in "$@"; do
    timeout -k 10 200 python bench.py --steps $S --spinup 3 --no-cpu-baseline $spec 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('[$spec]', d['ms_per_step'], d['value'], flush=True)
"
  done
done
