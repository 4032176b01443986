#!/bin/bash
# Dev tool: A/B several builds of libudapose_hip.so (tools/_ab/lib_<tag>.so) on the same box, interleaved.
# usage: tools/ab_libs.sh "<tag> <tag> ..." [rounds] [steps]
TAGS="$1"; R="${2:-2}"; S="${3:-60}"
L=uda_poseestimation_amd/libudapose_hip.so
cp $L /tmp/lib_keep.so
# (the production library is restored however the script ends - interrupt, timeout kill, failed cp)
trap 'cp /tmp/lib_keep.so $L' EXIT
for i in $(seq 1 $R); do
  for tag in $TAGS; do
    cp tools/_ab/lib_$tag.so $L
    timeout -k 10 200 python bench.py --steps $S --spinup 4 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$tag', d['ms_per_step'], d['value'])
"
  done
done
cp /tmp/lib_keep.so $L
