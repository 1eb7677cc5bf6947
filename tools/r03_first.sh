#!/bin/bash
# round 3, first GPU contact: new-format parity tests, then the variable-coefficient 512^3 product in formats 5 / 1 / 0
# and a grid sweep of format 5
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_varcoef.py tests/test_gpu_formats.py -x -q 2>&1 | tail -15 | tee gpurun_out/r03_first_tests.txt
B="python bench.py --no-cpu --no-extra --steps 100 --warmup 10"
MK_DEBUG_PLAN=1 timeout 600 $B > gpurun_out/vc_fmt5.json 2> gpurun_out/vc_fmt5.err
for f in 1 0; do MK_SPMV_FORMAT=$f timeout 600 $B > gpurun_out/vc_fmt$f.json 2> gpurun_out/vc_fmt$f.err; done
for g in 1024 1280 1536 2048; do MK_GRID_SPMV=$g timeout 600 $B > gpurun_out/vc_fmt5_g$g.json 2> gpurun_out/vc_fmt5_g$g.err; done
for m in 0 1; do MK_SPMV_MAP=$m timeout 600 $B > gpurun_out/vc_fmt5_map$m.json 2> gpurun_out/vc_fmt5_map$m.err; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/vc_fmt*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d['roofline']
        print(f, 'its %.1f' % d['value'], 'spmv_us %.1f' % r['avg_launch_us'], 'frac %.3f' % r['frac'], 'fmt', d['config']['storage_format']['format'], 'grid', d['config']['storage_format']['grid'])
    except Exception as e:
        print(f, 'FAILED', e)
PY
tail -3 gpurun_out/vc_fmt5.err
