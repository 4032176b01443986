# Round-6 profile collection (one gpurun call): bench lines, kernel stats, HBM traffic PMC passes, SQ counters.
export TMPDIR=/tmp
O=gpurun_out/r6prof; mkdir -p $O
python3 bench.py --steps 20 --warmup 3 > $O/r6_bench_steps20.json 2> $O/r6_bench_steps20.err
python3 bench.py > $O/r6_bench_default.json 2> $O/r6_bench_default.err
python3 bench.py --steps 60 --no-cpu-baseline --precision reference > $O/r6_bench_reference.json 2>/dev/null
python3 bench.py --steps 60 --no-cpu-baseline --dtype fp16 > $O/r6_bench_config1_fp16.json 2>/dev/null
python3 bench.py --steps 30 --no-cpu-baseline --image-size 384 --keypoints 18 --sigma 1.0 --dtype fp16 > $O/r6_bench_config4_fp16.json 2>/dev/null
python3 bench.py --steps 40 --no-cpu-baseline --config2 > $O/r6_bench_config2_bf16style.json 2>/dev/null
python3 bench.py --steps 40 --no-cpu-baseline --config2 --precision reference > $O/r6_bench_config2_reference.json 2>/dev/null
echo "bench lines done"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --spinup 4 --no-cpu-baseline --no-other-configs > $GRAFT_REPO_ROOT/$O/r6_bench_rocprof_stdout.txt 2> $GRAFT_REPO_ROOT/$O/r6_prof_stderr.txt
echo "kernel-trace rc=$?"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --eager --steps 3 --warmup 1 --spinup 0 --no-cpu-baseline > /dev/null 2>&1
echo "pmc fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py --eager --steps 3 --warmup 1 --spinup 0 --no-cpu-baseline > /dev/null 2>&1
echo "pmc write rc=$?"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_sq -- python3 $GRAFT_REPO_ROOT/bench.py --eager --steps 2 --warmup 1 --spinup 0 --no-cpu-baseline > /dev/null 2>&1
echo "pmc sq rc=$?"
cd $GRAFT_REPO_ROOT
python3 tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/r6_pmc_hbm_traffic.txt > /dev/null
python3 tools/pmc_by_shape.py $O/pmc_fetch $O/pmc_write $O/r6_pmc_by_shape.txt 9 | tail -1
python3 tools/pmc_sq_by_shape.py $O/pmc_sq $O/r6_pmc_sq_by_shape.txt 16 > /dev/null
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/r6_bench_kernel_stats.csv
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
du -sh $O; ls $O
