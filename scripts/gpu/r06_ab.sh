#!/bin/bash
# round 6, call ab: the statistics tests with the in-between sizes
O=gpurun_out/r06ab; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "quality_stats or long_reads" -s > $O/pytest_stats.txt 2>&1; tail -n 5 $O/pytest_stats.txt | cut -c1-300
