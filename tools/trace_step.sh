# Dev tool: kernel trace of a short graph-replayed run, analysed on the box (the trace itself is not merged back)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_step -- python3 bench.py --steps 6 --warmup 2 --spinup 2 --no-cpu-baseline "$@" > gpurun_out/trace_stdout.txt 2> gpurun_out/trace_stderr.txt
echo "rc=$?"
f=$(find /tmp/trace_step -name "*kernel_trace.csv" | head -1)
head -1 "$f" | cut -c1-400
python3 tools/timeline.py "$f" 3
