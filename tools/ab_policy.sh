#!/bin/bash
# Whole-step A/B of dispatch-policy values on ONE box, alternating: tools/ab_policy.sh FIELD v0 v1 v2 ...   (ms/step of bench.py --steps 60)
F=$1; shift
for i in 1 2; do
  for v in "$@"; do
    python bench.py --steps 60 --warmup 5 --spinup 4 --no-other-configs --no-cpu-baseline --policy $F=$v 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$F=$v'.ljust(24), d['ms_per_step'], d['value'], 'igemm', d['roofline']['kernel_ms_per_step'], d['roofline']['avg_launch_us'])"
  done
done
