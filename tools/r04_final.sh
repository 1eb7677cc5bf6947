#!/bin/bash
O=gpurun_out/r4i; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/all.txt 2>&1; tail -3 $O/all.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
( time python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench_time.txt; tail -3 $O/bench_time.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4i/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['roofline']['traffic'] and d['roofline']['traffic']['bytes'])
print(d['residual']['rel_gap'], d['placement_draws']['count'], d['cpu_baseline']['value'], d['cpu_baseline']['sample_rows'])
print(d['config_literal']['value'], d['config_literal']['roofline']['frac'])
for k,v in d['solver_loops'].items(): print(k, round(v['value'],1), round(v['iteration_frac'],3), v.get('product_us'))
PY
