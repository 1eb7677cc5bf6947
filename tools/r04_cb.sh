#!/bin/bash
O=gpurun_out/r4d; mkdir -p $O
export BENCH_ONLY_LOOPS=lsqr
for kb in auto 1024 2048 3072; do
  if [ $kb = auto ]; then unset MK_COLBLOCK_KB; else export MK_COLBLOCK_KB=$kb; fi
  python bench.py --only-other-configs > $O/cb_$kb.json 2> $O/cb_$kb.err
  python - <<PY
import json
d=json.loads(open('$O/cb_$kb.json').read().strip().splitlines()[-1])
for k,e in d.items():
    print('colblock_kb=$kb', k, round(e['value'],1), {a: round(b['avg_product_us'],1) for a,b in e['products'].items()}, e['format']['format'], e['format_transpose']['format'])
PY
done
