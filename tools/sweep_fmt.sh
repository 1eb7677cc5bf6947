#!/bin/bash
# sweep SpMV format x grid cap x tile order on the GPU box: prints value / SpMV us for 512^3 and 2-D n=1e6
R=${GRAFT_REPO_ROOT:-$(pwd)}
for fmt in ${FMTS:-1 2}; do for grid in ${GRIDS:-1024 1280 1536 2048}; do for map in ${MAPS:-0 1 2}; do
  MK_SPMV_FORMAT=$fmt MK_GRID_SPMV=$grid MK_SPMV_MAP=$map python $R/bench.py --steps 200 --warmup 20 --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
e=d['extra']['poisson2d-1000@1']
print('fmt=$fmt grid=$grid map=$map : 3d %.1f it/s spmv %.1f us | 2d %.0f it/s spmv %.2f us' % (d['value'], d['roofline']['avg_launch_us'], e['value'], e['roofline']['avg_launch_us']))"
done; done; done
