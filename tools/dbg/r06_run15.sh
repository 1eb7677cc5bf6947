cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_march_general.py tests/test_gpu_pencil.py -q -x 2>&1 | tail -4 > gpurun_out/r06o_tests.txt
cat gpurun_out/r06o_tests.txt
STEPS=300 timeout 900 python tools/r06_march_sizes.py 480,480,480 496,496,496 500,500,500 504,504,504 512,512,512 528,528,528 2>&1 | cut -c1-260 | sed 's/.*|  fmt/   | fmt/' > gpurun_out/r06o_align.txt
cat gpurun_out/r06o_align.txt
