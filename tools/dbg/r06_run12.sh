cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_slab_march.py tests/test_gpu_full_size.py -q -x 2>&1 | tail -15 > gpurun_out/r06l_tests.txt
cat gpurun_out/r06l_tests.txt
