#!/bin/bash
O=gpurun_out/r4b; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_dist.py -x -q -m gpu > $O/tests.txt 2>&1
tail -25 $O/tests.txt
