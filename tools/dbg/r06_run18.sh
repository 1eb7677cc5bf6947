cd $GRAFT_REPO_ROOT
for i in 1 2; do
( time timeout 900 python bench.py ) > gpurun_out/r06r_bench_$i.json 2> gpurun_out/r06r_bench_$i.err
cp bench_detail.json gpurun_out/r06r_bench_detail_$i.json
python -c "
import json; l=json.loads(open('gpurun_out/r06r_bench_$i.json').read().strip().splitlines()[-1]); print(l['value'], l['roofline']['frac'], l['roofline'].get('traffic_ratio'), l['second_workload']['value'], l['csr_plain'])"
done
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r06r_bench_driver.json 2> gpurun_out/r06r_bench_driver.err
