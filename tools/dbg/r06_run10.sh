cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06j_smoke.txt 2>&1; tail -2 gpurun_out/r06j_smoke.txt
( time timeout 900 python bench.py ) > gpurun_out/r06j_bench.json 2> gpurun_out/r06j_bench.err
cp bench_detail.json gpurun_out/r06j_bench_detail.json
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r06j_bench_driver.json 2> gpurun_out/r06j_bench_driver.err
tail -c 300 gpurun_out/r06j_bench.json
( time timeout 3000 bash tools/profile_r06.sh r06b ) > gpurun_out/r06j_profile.log 2>&1
grep -A8 "poisson3d-512@1" gpurun_out/prof_r06b/spmv_traffic.json | head -12
timeout 600 python -m pytest tests/test_gpu_march_general.py tests/test_gpu_pencil.py tests/test_gpu_slab_march.py -q 2>&1 | tail -3
