#!/bin/bash
# round 4, call i: the whole GPU tier (with the matrix) at the current tree
mkdir -p gpurun_out/r04i
timeout 2400 python -m pytest tests -q -s -m gpu -x > gpurun_out/r04i/pytest_gpu.txt 2>&1
grep "^waves\|passed\|failed\|Error" gpurun_out/r04i/pytest_gpu.txt | tail -20
