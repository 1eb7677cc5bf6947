#!/bin/bash
# A/B of two builds of libmikrylov.so on the SAME box: tools/ab/run.sh libA.so libB.so [bench args]
R=${GRAFT_REPO_ROOT:-$(pwd)}
A=$1; B=$2; shift 2
cp $R/pykrylov_amd/libmikrylov.so /tmp/lib_orig.so
for rep in 1 2; do for L in $A $B; do
  cp $R/tools/ab/$L $R/pykrylov_amd/libmikrylov.so
  python $R/bench.py --steps 300 --warmup 20 --no-cpu "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d.get('extra',{}).get('poisson2d-1000@1')
print('$L rep$rep: 3d %.1f it/s spmv %.1f us' % (d['value'], d['roofline']['avg_launch_us']) + (' | 2d %.0f it/s spmv %.2f us' % (e['value'], e['roofline']['avg_launch_us']) if e else ''))"
done; done
cp /tmp/lib_orig.so $R/pykrylov_amd/libmikrylov.so
