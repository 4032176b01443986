O=gpurun_out/r5lines; mkdir -p $O
python3 bench.py --steps 20 --warmup 3 > $O/r5_bench_steps20.json 2> $O/r5_bench_steps20.err
python3 bench.py > $O/r5_bench_default.json 2> $O/r5_bench_default.err
python3 bench.py --steps 60 --no-cpu-baseline --precision reference > $O/r5_bench_reference.json 2>/dev/null
python3 bench.py --steps 60 --no-cpu-baseline --dtype fp16 > $O/r5_bench_config1_fp16.json 2>/dev/null
python3 bench.py --steps 30 --no-cpu-baseline --image-size 384 --keypoints 18 --sigma 1.0 --dtype fp16 > $O/r5_bench_config4_fp16.json 2>/dev/null
python3 bench.py --steps 40 --no-cpu-baseline --config2 > $O/r5_bench_config2_bf16style.json 2>/dev/null
python3 bench.py --steps 40 --no-cpu-baseline --config2 --precision reference > $O/r5_bench_config2_reference.json 2>/dev/null
for f in $O/*.json; do python3 - "$f" <<'P'
import sys, json
d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1]); print(sys.argv[1], d["value"], d["ms_per_step"])
P
done
