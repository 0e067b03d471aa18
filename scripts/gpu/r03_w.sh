#!/bin/bash
# round-3 GPU call W: clip instances after the 64-column fix and the 16-column form for reads beyond 255 bases: parity tests, then times
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r03w; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fuzz or configs or variable or long_reads or clip or adversarial" > $O/pytest_clip.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_clip.log
timeout 500 python scripts/debug/clip_k_case.py 2>&1 | grep -v "^   " | tail -5
timeout 600 python scripts/clip_by_adapter_len.py 13 17 20 24 28 32 34 40 48 64 > $O/clip_by_adapter_len_100.txt 2> $O/err.txt; cut -c1-150 $O/clip_by_adapter_len_100.txt
L=150 timeout 600 python scripts/clip_by_adapter_len.py 13 20 24 34 48 64 > $O/clip_by_adapter_len_150.txt 2>> $O/err.txt; cut -c1-150 $O/clip_by_adapter_len_150.txt
L=250 READS=10000000 timeout 600 python scripts/clip_by_adapter_len.py 13 20 34 64 > $O/clip_by_adapter_len_250.txt 2>> $O/err.txt; cut -c1-150 $O/clip_by_adapter_len_250.txt
L=300 READS=10000000 timeout 600 python scripts/clip_by_adapter_len.py 8 13 20 34 64 > $O/clip_by_adapter_len_300.txt 2>> $O/err.txt; cut -c1-150 $O/clip_by_adapter_len_300.txt
tail -3 $O/err.txt
