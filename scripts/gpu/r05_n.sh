#!/bin/bash
# round 5, call n: the scan time-out recovered without the scanner, the 56 / 80 column clip buckets, the rank path (stdout clean), pipes with larger pipes
O=gpurun_out/r05n; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "scan_timeout or adversarial or two_ranks" > $O/pytest_a.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_cli.py -q -x -m gpu -k "one_output_file or survives" > $O/pytest_b.txt 2>&1
timeout 300 python scripts/e2e_pipe.py > $O/e2e_pipe.txt 2>&1
READS=10000000 timeout 600 python scripts/clip_by_adapter_len.py 40 44 48 49 52 56 57 60 64 65 72 80 81 90 99 > $O/clip_by_adapter_len.txt 2>&1
WITH_N=1 READS=10000000 L=150 timeout 600 python scripts/clip_by_adapter_len.py 48 49 56 57 64 >> $O/clip_by_adapter_len.txt 2>&1
tail -8 $O/pytest_a.txt; tail -8 $O/pytest_b.txt; grep -v "timing part" $O/e2e_pipe.txt; tail -40 $O/clip_by_adapter_len.txt
