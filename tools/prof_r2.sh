export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r2 -- python3 bench.py --steps 20 --warmup 3 --spinup 4 --no-cpu-baseline > gpurun_out/r2_prof_stdout.txt 2> gpurun_out/r2_prof_stderr.txt
echo "kernel-trace rc=$?"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch_r2 -- python3 bench.py --eager --steps 3 --warmup 1 --spinup 0 --no-cpu-baseline > gpurun_out/r2_pmcf_stdout.txt 2> gpurun_out/r2_pmcf_stderr.txt
echo "pmc fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write_r2 -- python3 bench.py --eager --steps 3 --warmup 1 --spinup 0 --no-cpu-baseline > gpurun_out/r2_pmcw_stdout.txt 2> gpurun_out/r2_pmcw_stderr.txt
echo "pmc write rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r2_cfg2 -- python3 bench.py --config2 --steps 5 --warmup 2 --spinup 0 --no-cpu-baseline > gpurun_out/r2_prof2_stdout.txt 2> gpurun_out/r2_prof2_stderr.txt
echo "cfg2 rc=$?"
find gpurun_out -name "*stats*csv" | head; du -sh gpurun_out/pmc_fetch_r2 gpurun_out/pmc_write_r2 gpurun_out/prof_r2 gpurun_out/prof_r2_cfg2
# keep the merge small: drop the per-dispatch traces of the stats runs, keep stats + the counter csvs
find gpurun_out/prof_r2 gpurun_out/prof_r2_cfg2 -name "*kernel_trace.csv" -delete
find gpurun_out/pmc_fetch_r2 gpurun_out/pmc_write_r2 -name "*kernel_trace.csv" -delete
find gpurun_out -name "*.db" -delete
du -sh gpurun_out
