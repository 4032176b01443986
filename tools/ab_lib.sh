#!/bin/bash
# Dev tool: A/B two builds of libudapose_hip.so (tools/_ab/libold.so, tools/_ab/libnew.so) on the same box, interleaved.
R="${1:-2}"; S="${2:-60}"
L=uda_poseestimation_amd/libudapose_hip.so
cp $L /tmp/lib_keep.so
# the production library is restored however the script ends (interrupt, timeout kill, failed cp)
trap 'cp /tmp/lib_keep.so $L' EXIT
for i in $(seq 1 $R); do
  for tag in old new; do
    cp tools/_ab/lib$tag.so $L
    timeout -k 10 200 python bench.py --steps $S --spinup 4 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$tag', d['ms_per_step'], d['value'], 'igemm', d['roofline']['achieved'], 'TF', d['roofline']['kernel_ms_per_step'], 'ms')
"
  done
done
cp /tmp/lib_keep.so $L
