# Round-2 measurement set (one box): the driver's command, the default run, configs[2], configs[4] in fp16 and bf16, PCIe-inclusive.
export TMPDIR=/tmp
python3 bench.py --steps 20 --warmup 3 > gpurun_out/r2_bench_steps20.json 2> gpurun_out/r2_bench_steps20.err; echo "steps20 rc=$?"
python3 bench.py > gpurun_out/r2_bench_default.json 2> gpurun_out/r2_bench_default.err; echo "default rc=$?"
python3 bench.py --config2 --steps 20 --no-cpu-baseline > gpurun_out/r2_bench_config2_captured.json 2>/dev/null; echo "config2 captured rc=$?"
python3 bench.py --config2 --eager --steps 20 --no-cpu-baseline > gpurun_out/r2_bench_config2.json 2>/dev/null; echo "config2 eager rc=$?"
python3 bench.py --image-size 384 --keypoints 18 --sigma 1.0 --dtype fp16 --steps 40 --no-cpu-baseline > gpurun_out/r2_bench_config4_fp16.json 2>/dev/null; echo "config4 fp16 rc=$?"
python3 bench.py --image-size 384 --keypoints 18 --sigma 1.0 --dtype bf16 --steps 40 --no-cpu-baseline > gpurun_out/r2_bench_config4_bf16.json 2>/dev/null; echo "config4 bf16 rc=$?"
python3 bench.py --dtype fp16 --steps 60 --no-cpu-baseline > gpurun_out/r2_bench_config1_fp16.json 2>/dev/null; echo "config1 fp16 rc=$?"
python3 bench.py --host-inputs --steps 60 --no-cpu-baseline > gpurun_out/r2_bench_hostinputs.json 2>/dev/null; echo "host inputs rc=$?"
python3 bench.py --arch pose_resnet50 --steps 60 --no-cpu-baseline > gpurun_out/r2_bench_r50.json 2>/dev/null; echo "r50 rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r2_cfg2b -- python3 bench.py --config2 --steps 5 --warmup 2 --spinup 0 --no-cpu-baseline > /dev/null 2>&1; echo "cfg2 prof rc=$?"
find gpurun_out/prof_r2_cfg2b -name "*kernel_trace.csv" -delete; find gpurun_out -name "*.db" -delete
for f in gpurun_out/r2_bench_*.json; do python3 -c "
import json,sys
d=json.loads([l for l in open('$f') if l.startswith('{')][0]); print('$f', d['ms_per_step'], d['value'], d['dtype'], d['roofline']['achieved'], d['roofline']['frac'])"; done
