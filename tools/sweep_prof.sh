#!/bin/bash
# per-kernel durations (rocprofv3 kernel trace) for a few grid caps; usage: sweep_prof.sh <workload> "<bench args>" caps...
WL=$1; BARGS=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
for cfg in "$@"; do
  gs=${cfg%%:*}; gt=${cfg##*:}
  rm -rf /tmp/sw_prof
  MK_GRID_SPMV=$gs MK_GRID_STREAM=$gt rocprofv3 --kernel-trace --stats -f csv -d /tmp/sw_prof -o s -- python $R/bench.py --workload $WL $BARGS --no-cpu --no-extra >/tmp/sw.json 2>/dev/null
  python - "$gs" "$gt" <<'PY'
import csv,glob,sys,json
rows=list(csv.DictReader(open(glob.glob('/tmp/sw_prof/**/*kernel_stats.csv',recursive=True)[0])))
d=json.load(open('/tmp/sw.json'))
out=[]
for r in rows:
    n=r['Name']
    for key in ('CgSpmvEpi','CgUpdateXR','CgUpdateP'):
        if key in n: out.append('%s=%.1fus'%(key,float(r['AverageNs'])/1e3))
print('spmv=%s stream=%s ms/step=%.4f  '%(sys.argv[1],sys.argv[2],d['ms_per_step'])+'  '.join(sorted(out)))
PY
done
