# Dev tool: A/B of two whole trees (this one and tools/_ab/oldtree = a git worktree of an older commit with its own built library), interleaved on one box.
R="${1:-3}"; S="${2:-60}"; X="${3:-}"     # rounds, steps, extra bench arguments
for i in $(seq 1 $R); do
  for tag in old new; do
    if [ $tag = old ]; then D=tools/_ab/oldtree; else D=.; fi
    (cd $D && timeout -k 10 200 python bench.py --steps $S --spinup 3 --no-cpu-baseline --no-other-configs $X 2>/dev/null) | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$tag', d['ms_per_step'], d['value'])
"
  done
done
