cd $GRAFT_REPO_ROOT
( time timeout 900 python bench.py ) > gpurun_out/r06k_bench.json 2> gpurun_out/r06k_bench.err
cp bench_detail.json gpurun_out/r06k_bench_detail.json
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r06k_bench_driver.json 2> gpurun_out/r06k_bench_driver.err
head -c 1500 gpurun_out/r06k_bench.json
