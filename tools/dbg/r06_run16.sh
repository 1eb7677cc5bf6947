cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_march_general.py tests/test_gpu_pencil.py tests/test_gpu_slab_march.py -q -x 2>&1 | tail -4 > gpurun_out/r06p_tests.txt
cat gpurun_out/r06p_tests.txt
STEPS=300 timeout 900 python tools/r06_march_sizes.py 480,480,480 500,500,500 v:500,500,500 504,504,504 384,300,201 2>&1 | cut -c1-260 | sed 's/.*|  fmt/   | fmt/' > gpurun_out/r06p_sizes.txt
cat gpurun_out/r06p_sizes.txt
