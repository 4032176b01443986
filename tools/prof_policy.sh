# Dev tool: rocprofv3 --kernel-trace --stats of a short bench run under policy overrides; prints the per-kernel summary head.  usage: bash tools/prof_policy.sh "<bench args>"
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/prof_policy; rm -rf $O; mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --spinup 1 --no-cpu-baseline --no-other-configs $1 > $O/stdout.txt 2> $O/stderr.txt
echo "rc=$?"
cd $GRAFT_REPO_ROOT
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete; find $O -name "*agent_info.csv" -delete
python3 - <<'P'
import csv, os
rows = list(csv.DictReader(open(os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/prof_policy/kernel_stats.csv")))
steps = [int(r["Calls"]) for r in rows if "opt_tail_k" in r["Name"]][0]
print("steps", steps)
for r in rows[:26]:
    print(f'{int(r["Calls"]) / steps:8.1f} /step  {float(r["AverageNs"]) / 1e3:8.2f} us  {int(r["TotalDurationNs"]) / steps / 1e6:7.3f} ms/step  {r["Name"][:110]}')
P
