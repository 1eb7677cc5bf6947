cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 2400 python -m pytest tests -m gpu -q --durations=15 ) > gpurun_out/r06q_gputests.log 2>&1
tail -25 gpurun_out/r06q_gputests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time timeout 3000 bash tools/profile_r06.sh r06c ) > gpurun_out/r06q_profile.log 2>&1
cp gpurun_out/prof_r06c/spmv_traffic.json profiles/spmv_traffic.json
( time timeout 900 python bench.py ) > gpurun_out/r06q_bench.json 2> gpurun_out/r06q_bench.err
cp bench_detail.json gpurun_out/r06q_bench_detail.json
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r06q_bench_driver.json 2> gpurun_out/r06q_bench_driver.err
head -c 1200 gpurun_out/r06q_bench.json; echo
timeout 900 bash tools/r06_slab_budget.sh > gpurun_out/r06q_slab.log 2>&1
grep -E "^==|CgFused|kernels per pass" gpurun_out/r06q_slab.log
