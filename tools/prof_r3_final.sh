# Round-3 final refresh of the configs[2] / style-network evidence (one gpurun call)
export TMPDIR=/tmp
O=gpurun_out/r3final; mkdir -p $O
python3 bench.py --steps 40 --no-cpu-baseline --config2 > $O/r3_bench_config2_bf16style.json 2>/dev/null
python3 bench.py --steps 40 --no-cpu-baseline --config2 --precision reference > $O/r3_bench_config2_reference.json 2>/dev/null
python3 bench.py --steps 60 --no-cpu-baseline --precision reference > $O/r3_bench_reference.json 2>/dev/null
echo "bench lines done"
python3 tools/time_style_layers.py 32 2>&1 | grep -v amdgpu > $O/r3_style_layers.txt
python3 tools/time_style_layers.py 32 -1 0 2>&1 | grep -v amdgpu > $O/r3_style_layers_igemm.txt
python3 tools/time_modes.py bf16,fp32,f16x2 2>&1 | grep -v amdgpu > $O/r3_modes_timing.txt
echo "timings done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bf16 -- python3 bench.py --config2 --steps 20 --warmup 3 --spinup 2 --no-cpu-baseline > /dev/null 2> $O/err1.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ref -- python3 bench.py --config2 --precision reference --steps 20 --warmup 3 --spinup 2 --no-cpu-baseline > /dev/null 2> $O/err2.txt
for t in bf16 ref; do f=$(find $O/$t -name "*kernel_stats.csv" | head -1); cp "$f" $O/r3_config2_${t}_kernel_stats.csv; done
rm -rf $O/bf16 $O/ref
ls $O
