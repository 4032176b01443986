# Dev tool: the kernels a captured step runs between its last forward convolution and its first backward kernel (the serial "loss section"), with durations,
# from a rocprofv3 kernel trace of a short bench run.  usage (inside one gpurun call): bash tools/loss_section.sh
export TMPDIR=/tmp
O=/tmp/loss_sec; rm -rf $O; mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --spinup 1 --no-cpu-baseline --no-other-configs > $O/stdout.txt 2> $O/stderr.txt
echo "rc=$?"
cd $GRAFT_REPO_ROOT
python3 - <<'P'
import csv, glob
f = glob.glob('/tmp/loss_sec/**/*kernel_trace.csv', recursive=True)[0]
rows = []
for d in csv.DictReader(open(f)):
    rows.append((int(d["Start_Timestamp"]), int(d["End_Timestamp"]), d["Kernel_Name"]))
rows.sort()
tails = [i for i, x in enumerate(rows) if "opt_tail_k" in x[2]]
lo, hi = tails[-3] + 1, tails[-2]          # one graph-replayed step of the timed region
step = rows[lo:hi + 1]
first_bwd = next(i for i, x in enumerate(step) if "ELb1E" in x[2] or "bn_bwd" in x[2] or "sqdiff_bwd" in x[2])
last_fwd = max(i for i, x in enumerate(step[:first_bwd]) if "igemm" in x[2])
print("step kernels", len(step), "last forward conv at", last_fwd, "first backward kernel at", first_bwd)
t0 = step[last_fwd][1]
for s, e, n in step[last_fwd:first_bwd + 3]:
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:6.1f}  {n[:120]}")
print("--- head of the step")
t0 = step[0][0]
for s, e, n in step[:14]:
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:6.1f}  {n[:120]}")
P
