#!/bin/bash
O=gpurun_out/r4n; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_bitexact_full.py -x -q -m gpu > $O/tests.txt 2>&1; tail -3 $O/tests.txt
export BENCH_ONLY_LOOPS=lsqr,craigmr
for v in whole stepped whole stepped; do
  case $v in
    whole) export MK_RT_STEPPED=0;;
    stepped) unset MK_RT_STEPPED;;
  esac
  python bench.py --only-other-configs > $O/st_$v.json 2> $O/st_$v.err
  python - <<PY
import json
d=json.loads(open('$O/st_$v.json').read().strip().splitlines()[-1])
for k,e in d.items():
    print('$v', k, round(e['value'],1), {a: round(b['avg_product_us'],1) for a,b in e['products'].items()})
PY
done
