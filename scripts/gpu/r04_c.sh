#!/bin/bash
# round 4, call c: the clip instance x launch-bounds matrix on the GPU (libraries built here: scripts/build_clip_matrix.py)
mkdir -p gpurun_out/r04c
timeout 2400 python -m pytest tests/test_gpu_clip_matrix.py -q -s -m gpu > gpurun_out/r04c/pytest_matrix.txt 2>&1
tail -40 gpurun_out/r04c/pytest_matrix.txt
