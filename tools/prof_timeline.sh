#!/bin/bash
# Dev tool: per-step device timeline of a bench configuration.  usage: tools/prof_timeline.sh <tag> "<bench args>"
export TMPDIR=/tmp
T="$1"; A="$2"; O=gpurun_out/tl_$T
rm -rf /tmp/tl_$T; mkdir -p gpurun_out
cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$T -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --spinup 2 --no-cpu-baseline $A > /tmp/tl_$T.json 2> /tmp/tl_$T.err
cd $GRAFT_REPO_ROOT
python3 -c "
import json
for l in open('/tmp/tl_$T.json'):
    if l.startswith('{'):
        d = json.loads(l); print('$T', d['ms_per_step'], d['value'])
"
python3 tools/step_timeline.py /tmp/tl_$T 30 > $O.txt 2>&1
cat $O.txt
