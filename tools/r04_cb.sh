#!/bin/bash
# round 4: A/B of the least-squares product paths (two alternating runs each): column blocks on the gather path vs as resident tiles
# (MK_CB_RESIDENT), the many-step `A v` as one launch vs one launch per step (MK_RT_STEPPED), one tile per step vs pairs (MK_RT_REG)
O=gpurun_out/r4_lls; mkdir -p $O
export BENCH_ONLY_LOOPS=lsqr,craigmr
for v in base cb_gather whole one_tile base cb_gather whole one_tile; do
  unset MK_CB_RESIDENT MK_RT_STEPPED MK_RT_REG
  case $v in
    cb_gather) export MK_CB_RESIDENT=0;;
    whole) export MK_RT_STEPPED=0;;
    one_tile) export MK_RT_REG=0;;
  esac
  python bench.py --only-other-configs > $O/$v.json 2> $O/$v.err
  python - <<PY
import json
d=json.loads(open('$O/$v.json').read().strip().splitlines()[-1])
for k,e in d.items():
    print('$v', k, round(e['value'],1), {a: round(b['avg_product_us'],1) for a,b in e['products'].items()})
PY
done
