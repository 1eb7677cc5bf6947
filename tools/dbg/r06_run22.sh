cd $GRAFT_REPO_ROOT
( time timeout 3000 bash tools/profile_r06.sh r06e ) > gpurun_out/r06v_profile.log 2>&1
python -c "
import json
t=json.load(open('gpurun_out/prof_r06e/spmv_traffic.json'))
for k,v in t.items():
    if isinstance(v,dict) and 'bytes' in v: print(k, v['bytes'], round(v['avg_us_in_trace'],1))
l=json.loads(open('gpurun_out/prof_r06e/bench_trace_default.json').read().strip().splitlines()[-1]); print('default under rocprof', l['value'], l['roofline']['avg_launch_us'])"
