cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_march_general.py tests/test_gpu_pencil.py tests/test_gpu_slab_march.py -q -x 2>&1 | tail -25 > gpurun_out/r06y_tests.txt
cat gpurun_out/r06y_tests.txt
