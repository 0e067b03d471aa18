#!/bin/bash
O=gpurun_out/r06u; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_clip_matrix.py -q -m gpu -s > $O/pytest_matrix.txt 2>&1; grep "waves\|passed\|failed" $O/pytest_matrix.txt | cut -c1-300
