cd $GRAFT_REPO_ROOT
MK_PEN_GEN=1 timeout 900 python -m pytest tests/test_gpu_pencil.py tests/test_gpu_slab_march.py tests/test_gpu_cg.py tests/test_gpu_full_size.py -q 2>&1 | tail -8
