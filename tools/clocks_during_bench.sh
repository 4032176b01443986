# Dev tool: GPU clocks / power / temperature sampled by rocm-smi while the headline bench runs (is the step running at the chip's top clocks?).
python3 bench.py --steps 400 --warmup 3 --no-cpu-baseline --no-other-configs > /tmp/clk_bench.json 2>/dev/null &
BP=$!
sleep 9
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -i "sclk\|mclk\|fclk\|power\|junction\|Temperature" | tr -s ' ' | cut -c1-120
  echo "--"
  sleep 1.5
done
wait $BP
python3 - <<'P'
import json
d=json.loads([l for l in open("/tmp/clk_bench.json").read().splitlines() if l.startswith("{")][-1]); print("bench", d["value"], d["ms_per_step"])
P
echo "idle:"; sleep 3; rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|mclk\|power" | tr -s ' ' | cut -c1-120
