#!/bin/bash
# placement experiments: 6 alternating processes per variant, headline workload, no draws
O=gpurun_out/r4e; mkdir -p $O
run() { # name, env...
  name=$1; shift
  env "$@" python bench.py --no-cpu --no-extra --no-parity --steps 300 --warmup 30 > $O/$name.json 2> $O/$name.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
    print('$name', round(d['value'],1), 'spmv_us', round(d['roofline']['avg_launch_us'],1), 'ms', round(d['ms_per_step'],3))
except Exception as e: print('$name', 'FAILED', e)
PY
}
for rep in 1 2 3 4 5 6; do
  run base_$rep X=1
  run arena_$rep BENCH_ARENA_VECTORS=4
  run free_$rep MK_FREE_CSR=1
  run both_$rep BENCH_ARENA_VECTORS=4 MK_FREE_CSR=1
done
