cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_cg.py tests/test_gpu_full_size.py tests/test_gpu_multirank.py -q 2>&1 | tail -4
timeout 600 python bench.py --no-cpu --no-extra --steps 300 --warmup 20 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l['value'], l['placement_draws'])"
