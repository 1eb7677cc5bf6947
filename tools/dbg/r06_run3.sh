cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1700 python -m pytest tests -m gpu -x -q ) > gpurun_out/r06c_gputests.log 2>&1
tail -5 gpurun_out/r06c_gputests.log
( time timeout 900 python bench.py ) > gpurun_out/r06c_bench.json 2> gpurun_out/r06c_bench.err
cp bench_detail.json gpurun_out/r06c_bench_detail.json
tail -c 600 gpurun_out/r06c_bench.json
timeout 900 bash tools/r06_slab_budget.sh > gpurun_out/r06c_slab.log 2>&1
tail -30 gpurun_out/r06c_slab.log
