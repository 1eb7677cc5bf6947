#!/bin/bash
# round 4, first GPU call: host facts, the new full-size tests, the default bench line
O=gpurun_out/r4a; mkdir -p $O
(free -g; nproc; grep -c processor /proc/cpuinfo; cat /proc/meminfo | head -3) > $O/host.txt 2>&1
echo skip tests > $O/tests.txt
tail -5 $O/tests.txt
timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"; tail -c 1500 $O/bench.err
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r4a/bench.json').read().strip().splitlines()[-1])
    print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['residual'], d['placement_draws'])
    print(json.dumps(d.get('cpu_baseline'))[:600])
    print(json.dumps(d.get('cpu_baseline_all_cores'))[:400])
    print(json.dumps(d.get('config_literal'))[:500])
    print(json.dumps(d.get('solver_loops'), indent=0)[:3000])
except Exception as e: print('parse failed', e)
PY
