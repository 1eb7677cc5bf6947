cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_march_general.py -q -x -k automatic 2>&1 | tail -15
