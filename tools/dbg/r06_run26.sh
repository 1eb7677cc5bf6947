cd $GRAFT_REPO_ROOT
( time MK_SPMV_NT=1 timeout 2400 python -m pytest tests -m gpu -q --deselect tests/test_gpu_full_size.py::test_non_temporal_accesses_are_the_default_beyond_the_infinity_cache ) > gpurun_out/r06x_nt_suite.log 2>&1
tail -6 gpurun_out/r06x_nt_suite.log
