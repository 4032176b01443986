#!/bin/bash
# Dev tool: A/B two bench configurations on the same box, interleaved (boxes differ by +-2 %, so compare within one call).
# usage: tools/ab.sh "<bench args A>" "<bench args B>" [rounds] [steps]
#   e.g. tools/ab.sh "" "--policy igemm_h3=0" 2 60        (dispatch-policy fields: include/udapose.h udapose_policy)
A="$1"; B="$2"; R="${3:-2}"; S="${4:-60}"
for i in $(seq 1 $R); do
  for cfg in "A:$A" "B:$B"; do
    tag="${cfg%%:*}"; spec="${cfg#*:}"
    timeout -k 10 200 python bench.py --steps $S --spinup 4 --no-cpu-baseline $spec 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$tag [$spec]', d['ms_per_step'], d['value'])
"
  done
done
