#!/bin/bash
# round 6, call o: the whole GPU tier (clip matrix included) at HEAD, then the closing evidence
O=gpurun_out/r06o; mkdir -p $O
timeout 3000 python -m pytest tests -x -q -m gpu --durations=10 > $O/pytest_gpu.txt 2>&1; tail -n 18 $O/pytest_gpu.txt
bash scripts/gpu/r06_final.sh 2>&1 | tail -40
