#!/bin/bash
# grid-cap sweep for the CG kernels (GPU box)
for wl in poisson2d-1000 poisson3d-512; do
  if [ $wl = poisson2d-1000 ]; then ST="--steps 2000 --warmup 200"; else ST="--steps 60 --warmup 6 --spmv-launches 20"; fi
  for gs in 512 1024 2048; do for gt in 512 1024 2048; do
    MK_GRID_SPMV=$gs MK_GRID_STREAM=$gt python bench.py --workload $wl $ST --no-cpu --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$wl spmv=$gs stream=$gt  it/s=%.1f  ms/step=%.4f  spmv_us=%.2f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_us']))"
  done; done
done
