#!/bin/bash
O=gpurun_out/r4l; mkdir -p $O
run() { name=$1; shift
  env "$@" python bench.py --no-cpu --no-extra --no-parity --steps 200 --warmup 20 > $O/$name.json 2> $O/$name.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
    print('$name', round(d['value'],1), 'spmv_us', round(d['roofline']['avg_launch_us'],1), 'rest_ms', round(d['ms_per_step']-d['roofline']['avg_launch_us']/1e3,3))
except Exception as e: print('$name', 'FAILED', e)
PY
}
for rep in 1 2 3 4 5 6; do
  run base_$rep X=1
  run gs256_$rep MK_GRID_STREAM=256
  run gs384_$rep MK_GRID_STREAM=384
done
