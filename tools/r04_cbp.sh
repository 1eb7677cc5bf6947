#!/bin/bash
# round 4: block size and phases of the column blocks of the lls A' u product (one run per setting)
O=gpurun_out/r4r; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_round4.py tests/test_gpu_lls_full_size.py tests/test_gpu_lls.py -x -q -m gpu > $O/tests.txt 2>&1; tail -2 $O/tests.txt
export BENCH_ONLY_LOOPS=lsqr
for v in "X=1" "MK_CB_PHASES=1" "MK_COLBLOCK_KB=4096" "MK_COLBLOCK_KB=6144" "MK_COLBLOCK_KB=8192" "MK_COLBLOCK_KB=11000" "X=1"; do
  env $v python bench.py --only-other-configs > $O/r.json 2> $O/r.err
  python - <<PY
import json
d=json.loads(open('$O/r.json').read().strip().splitlines()[-1])
for k,e in d.items():
    print('$v', round(e['value'],1), {a: round(b['avg_product_us'],1) for a,b in e['products'].items()})
PY
done
