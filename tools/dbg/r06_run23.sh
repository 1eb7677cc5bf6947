cd $GRAFT_REPO_ROOT
MK_SPMV_NT=1 timeout 900 python -m pytest tests/test_gpu_march_general.py tests/test_gpu_slab_march.py tests/test_gpu_pencil.py -q -x 2>&1 | tail -4
MK_PEN_GEN=1 timeout 900 python -m pytest tests/test_gpu_pencil.py tests/test_gpu_slab_march.py tests/test_gpu_cg.py -q -x 2>&1 | tail -4
MK_PEN_TAIL_GEN=0 timeout 900 python -m pytest tests/test_gpu_slab_march.py -q -x 2>&1 | tail -3
