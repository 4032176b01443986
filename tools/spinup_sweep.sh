# Dev tool: headline bench (--steps 20 --warmup 5) against the length of the untimed spin-up, interleaved on one box.
for r in 1 2; do
  for s in 0 1 3 8 15; do
    timeout -k 10 200 python bench.py --steps 20 --warmup 5 --spinup $s --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('spinup $s', d['ms_per_step'], d['value'], 'synced', d['ms_per_step_synced'])
"
  done
done
