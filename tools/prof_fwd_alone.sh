# Dev tool: per-kernel durations of ONE stream's eager forward / forward + backward (no other stream beside it): rocprofv3 --kernel-trace --stats over
# tools/time_fwd_policy.py.  usage (inside one gpurun call): bash tools/prof_fwd_alone.sh
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/fwd_alone; mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $GRAFT_REPO_ROOT/tools/time_fwd_policy.py "" > $O/stdout.txt 2> $O/stderr.txt
echo "rc=$?"
cd $GRAFT_REPO_ROOT
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/fwd_alone_kernel_stats.csv
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete; find $O -name "*agent_info.csv" -delete
cat $O/stdout.txt
head -25 $O/fwd_alone_kernel_stats.csv | cut -c1-200
