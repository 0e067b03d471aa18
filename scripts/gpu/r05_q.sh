#!/bin/bash
# round 5, call q: the rest of the GPU tier (from the launch-bounds matrix on)
O=gpurun_out/r05q; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_clip_matrix.py tests/test_gpu_parity.py tests/test_gpu_text.py -q -m gpu > $O/pytest_gpu_rest.txt 2>&1
tail -15 $O/pytest_gpu_rest.txt
