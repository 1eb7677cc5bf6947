cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 2400 python -m pytest tests -m gpu -q ) > gpurun_out/r06w_gputests.log 2>&1
tail -4 gpurun_out/r06w_gputests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time timeout 3000 bash tools/profile_r06.sh r06f ) > gpurun_out/r06w_profile.log 2>&1
cp gpurun_out/prof_r06f/spmv_traffic.json profiles/spmv_traffic.json
( time timeout 900 python bench.py ) > gpurun_out/r06w_bench.json 2> gpurun_out/r06w_bench.err
cp bench_detail.json gpurun_out/r06w_bench_detail.json
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r06w_bench_driver.json 2> gpurun_out/r06w_bench_driver.err
python -c "
import json
t=json.load(open('gpurun_out/prof_r06f/spmv_traffic.json'))
for k,v in t.items():
    if isinstance(v,dict) and 'bytes' in v: print(k, v['bytes'], round(v['avg_us_in_trace'],1))
l=json.loads(open('gpurun_out/prof_r06f/bench_trace_default.json').read().strip().splitlines()[-1]); print('default under rocprof', l['value'], l['roofline']['avg_launch_us'])
for f in ('r06w_bench','r06w_bench_driver'):
    l=json.loads(open('gpurun_out/%s.json'%f).read().strip().splitlines()[-1]); print(f, l['value'], l['ms_per_step'], l['roofline']['frac'], l['roofline'].get('traffic_ratio'), l['second_workload']['value'], l['build_sha'])"
