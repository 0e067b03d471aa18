#!/bin/bash
# round-3 GPU call V: clip kernel per adapter length with the two-pass form for 17..99 bases (and its one-pass form, FXG_CLIP_K_ONE_PASS=1); parity tests of the clip instances
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r03v; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fuzz or configs or variable or long_reads or clip or adversarial" > $O/pytest_clip.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_clip.log
timeout 600 python scripts/clip_by_adapter_len.py 13 17 20 24 28 32 34 40 48 64 99 > $O/clip_by_adapter_len_100.txt 2> $O/err.txt; cat $O/clip_by_adapter_len_100.txt | cut -c1-150
FXG_CLIP_K_ONE_PASS=1 timeout 600 python scripts/clip_by_adapter_len.py 20 34 64 > $O/clip_by_adapter_len_100_one_pass.txt 2>> $O/err.txt; cat $O/clip_by_adapter_len_100_one_pass.txt | cut -c1-150
L=150 timeout 600 python scripts/clip_by_adapter_len.py 13 20 24 34 48 64 > $O/clip_by_adapter_len_150.txt 2>> $O/err.txt; cat $O/clip_by_adapter_len_150.txt | cut -c1-150
L=250 READS=10000000 timeout 600 python scripts/clip_by_adapter_len.py 13 20 34 64 > $O/clip_by_adapter_len_250.txt 2>> $O/err.txt; cat $O/clip_by_adapter_len_250.txt | cut -c1-150
L=300 READS=10000000 timeout 600 python scripts/clip_by_adapter_len.py 13 20 34 64 > $O/clip_by_adapter_len_300.txt 2>> $O/err.txt; cat $O/clip_by_adapter_len_300.txt | cut -c1-150
tail -3 $O/err.txt
