#!/bin/bash
# experiment sweep: tile order 3 (XCD stripes), non-temporal accesses, value prefetch.  Variant libraries are built with
#   python -m pykrylov_amd.build --tag <tag> -D<MACRO> ...      and selected through MIKRYLOV_LIB.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/var
L=$PWD/pykrylov_amd
run() {   # name, env..., -- args
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --no-cpu --no-extra --steps 300 --warmup 30 "$@" > gpurun_out/var/$name.json 2> gpurun_out/var/$name.err
}
for wl in poisson3d-512-varcoef poisson3d-512; do
  s=${wl#poisson3d-512}; s=${s:-const}
  run base_$s X=1 -- --workload $wl
  for st in 32 64 128 256 512; do run map3_s${st}_$s MK_SPMV_MAP=3 MK_SPMV_STRIPE=$st -- --workload $wl; done
done
for t in ntls ntl nts f5nt f5pf f5ntpf; do
  run lib_$t MIKRYLOV_LIB=$L/libmikrylov_$t.so -- --workload poisson3d-512-varcoef
done
for t in f5nt f5pf f5ntpf; do
  run lib_${t}_map3 MIKRYLOV_LIB=$L/libmikrylov_$t.so MK_SPMV_MAP=3 MK_SPMV_STRIPE=128 -- --workload poisson3d-512-varcoef
done
for g in 1024 2048; do
  run gs${g} MK_GRID_STREAM=$g -- --workload poisson3d-512-varcoef
  run gs${g}_ntls MK_GRID_STREAM=$g MIKRYLOV_LIB=$L/libmikrylov_ntls.so -- --workload poisson3d-512-varcoef
done
python - <<'PY'
import json, glob, os
rows = []
for f in sorted(glob.glob('gpurun_out/var/*.json'), key=os.path.getmtime):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d['roofline']
        rows.append('%-28s its %7.1f  step %7.3f ms  spmv %7.1f us  rest %7.3f ms  frac %.3f fmt %d grid %d map %d' % (
            os.path.basename(f)[:-5], d['value'], d['ms_per_step'], r['avg_launch_us'], d['ms_per_step'] - r['avg_launch_us'] / 1e3,
            r['frac'], d['config']['storage_format']['format'], d['config']['storage_format']['grid'], d['config']['storage_format']['tile_order']))
    except Exception as e:
        rows.append('%s FAILED %r' % (f, e))
open('gpurun_out/var/summary.txt', 'w').write('\n'.join(rows) + '\n')
print('\n'.join(rows))
PY
