#!/bin/bash
O=gpurun_out/r4f; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_anchors.py -x -q -m gpu -s > $O/anchors.txt 2>&1; grep -E "passes|products|passed|failed|Error|assert" $O/anchors.txt | tail -20
timeout 2400 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_anchors.py > $O/all.txt 2>&1; tail -5 $O/all.txt
