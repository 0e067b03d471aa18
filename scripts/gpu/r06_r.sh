#!/bin/bash
# round 6, call r: the launch-bounds matrix with refused instances left unlaunched, then the rest of the GPU tier
O=gpurun_out/r06r; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_clip_matrix.py -q -m gpu -s > $O/pytest_matrix.txt 2>&1; tail -n 8 $O/pytest_matrix.txt | cut -c1-300
timeout 3000 python -m pytest tests -q -m gpu --deselect tests/test_gpu_clip_matrix.py > $O/pytest_gpu_rest.txt 2>&1; tail -n 4 $O/pytest_gpu_rest.txt
