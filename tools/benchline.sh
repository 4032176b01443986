#!/bin/bash
# Dev tool: run bench.py with the given args, print "<tag> ms_per_step value" (+ the spin-up trace)
tag="$1"; shift
timeout 180 python bench.py --no-cpu-baseline "$@" 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$tag', d['ms_per_step'], d['value'])
    elif l.startswith('spin-up'):
        print('$tag', l.strip())
"
