#!/bin/bash
# round 5, call o: the recovered scan time-out (re-planned decision-only pass), rank path with a clean stdout
O=gpurun_out/r05o; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "scan_timeout or two_ranks" > $O/pytest_a.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_cli.py -q -x -m gpu -k "one_output_file or survives" > $O/pytest_b.txt 2>&1
tail -12 $O/pytest_a.txt; tail -12 $O/pytest_b.txt
