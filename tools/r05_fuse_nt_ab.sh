#!/bin/bash
# round 5: non-temporal accesses for the fused CG kernel's own streams (x loads, x stores, the new p's stores), variant libraries
# built with `python -m pykrylov_amd.build --tag nt<k> -DMK_FUSE_NT_DEF=<k>` (1 x loads, 2 x stores, 4 p stores), alternating
# processes on one box, placement draws off so that every process keeps the state it was born with
cd "$GRAFT_REPO_ROOT" || exit 1
L=$PWD/pykrylov_amd
for rep in 1 2 3; do
  for t in base nt7 nt6 nt1; do
    lib=$L/libmikrylov.so; [ $t != base ] && lib=$L/libmikrylov_$t.so
    for wl in poisson3d-512 poisson3d-512-varcoef; do
      MK_PLACEMENT_DRAWS=1 MIKRYLOV_LIB=$lib python bench.py --workload $wl --no-extra --no-cpu --no-parity --steps 200 --warmup 20 2>/dev/null | tail -1 | python -c "
import sys,json
l=json.loads(sys.stdin.read())
print('$t', '$wl', round(l['value'],1), l['roofline']['avg_launch_us'])
"
    done
  done
done
