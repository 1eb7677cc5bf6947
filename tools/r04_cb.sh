#!/bin/bash
O=gpurun_out/r4o; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_lls_full_size.py tests/test_gpu_lls.py -x -q -m gpu > $O/tests.txt 2>&1; tail -3 $O/tests.txt
export BENCH_ONLY_LOOPS=lsqr,craigmr
for v in launches fused launches fused; do
  case $v in
    launches) export MK_CB_FUSED=0;;
    fused) unset MK_CB_FUSED;;
  esac
  python bench.py --only-other-configs > $O/cb_$v.json 2> $O/cb_$v.err
  python - <<PY
import json
d=json.loads(open('$O/cb_$v.json').read().strip().splitlines()[-1])
for k,e in d.items():
    print('$v', k, round(e['value'],1), {a: round(b['avg_product_us'],1) for a,b in e['products'].items()})
PY
done
