# Dev tool: kernel statistics of the configs[2] step (style transfer + occlusion), bf16 style and the reference precision mix.
export TMPDIR=/tmp
O=gpurun_out/r3c2; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bf16 -- python3 bench.py --config2 --steps 20 --warmup 3 --spinup 2 --no-cpu-baseline > $O/r3_config2_bf16style_stdout.txt 2> $O/err1.txt
echo "bf16 rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ref -- python3 bench.py --config2 --precision reference --steps 20 --warmup 3 --spinup 2 --no-cpu-baseline > $O/r3_config2_reference_stdout.txt 2> $O/err2.txt
echo "ref rc=$?"
for t in bf16 ref; do f=$(find $O/$t -name "*kernel_stats.csv" | head -1); cp "$f" $O/r3_config2_${t}_kernel_stats.csv; done
rm -rf $O/bf16 $O/ref
ls -la $O
