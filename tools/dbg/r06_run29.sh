cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_march_general.py tests/test_gpu_pencil.py tests/test_gpu_slab_march.py -q 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
