cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_march_general.py -x -q 2>&1 | tail -40 > gpurun_out/r06a_general.log
timeout 900 python -m pytest tests/test_gpu_pencil.py tests/test_gpu_slab_march.py -x -q 2>&1 | tail -30 > gpurun_out/r06a_pencil.log
