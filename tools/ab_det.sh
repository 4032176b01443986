#!/bin/bash
# A/B of the deterministic weight-gradient split reductions inside the step (same box, alternating): ms/step of the driver's command
for i in 1 2; do
  for pol in "" "--policy wgrad_det=0"; do
    python bench.py --steps 60 --warmup 5 --spinup 4 --no-other-configs --no-cpu-baseline $pol 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$pol'.ljust(22) or 'default', d['ms_per_step'], d['value'], d['roofline']['wgrad'])"
  done
done
