cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 2400 python -m pytest tests -m gpu -q --durations=12 ) > gpurun_out/r06u_gputests.log 2>&1
tail -18 gpurun_out/r06u_gputests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
STEPS=400 timeout 600 python tools/r06_march_sizes.py 2000,0,0 3000,0,0 4000,0,0 2>&1 | cut -c1-260 > gpurun_out/r06u_2d.txt; cat gpurun_out/r06u_2d.txt
( time timeout 3000 bash tools/profile_r06.sh r06d ) > gpurun_out/r06u_profile.log 2>&1
cp gpurun_out/prof_r06d/spmv_traffic.json profiles/spmv_traffic.json
( time timeout 900 python bench.py ) > gpurun_out/r06u_bench.json 2> gpurun_out/r06u_bench.err
cp bench_detail.json gpurun_out/r06u_bench_detail.json
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r06u_bench_driver.json 2> gpurun_out/r06u_bench_driver.err
python -c "
import json
for f in ('r06u_bench','r06u_bench_driver'):
    l=json.loads(open('gpurun_out/%s.json'%f).read().strip().splitlines()[-1]); print(f, l['value'], l['ms_per_step'], l['roofline']['frac'], l['roofline'].get('traffic_ratio'), l['second_workload']['value'], l['build_sha'])"
