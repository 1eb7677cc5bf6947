#!/bin/bash
# Round 6: what differs between a "fast" and a "slow" box?  Clocks, power and temperature sampled WHILE the headline loop runs
# (CG 512^3, 4000 passes), next to the loop's own rate.   bash tools/r06_box_state.sh > gpurun_out/box_state.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python bench.py --workload poisson3d-512 --no-cpu --no-extra --no-parity --steps 4000 --warmup 50 > /tmp/box_bench.json 2> /tmp/box_bench.err &
BP=$!
sleep 7
for i in 1 2 3; do
  timeout 20 rocm-smi --showclocks --showpower --showtemp --showperflevel 2>&1 | grep -E "sclk|mclk|fclk|socclk|Power|Temperature \(Sensor (junction|memory)|Performance Level" | sed 's/^/   /'
  echo "   --"
  sleep 1
done
wait $BP
python - <<'PY'
import json
l = json.loads(open("/tmp/box_bench.json").read().strip().splitlines()[-1])
print("loop: %.1f it/s, fused product %.1f us (%.3f of 8 TB/s), draws %s" % (l["value"], l["roofline"]["avg_launch_us"], l["roofline"]["frac"], l["placement_draws"]))
PY
timeout 20 rocm-smi --showproductname --showserial 2>&1 | grep -E "Card|Serial|SKU|GFX" | head -6
