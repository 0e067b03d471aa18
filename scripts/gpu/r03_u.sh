#!/bin/bash
# round-3 GPU call U: the clip kernel per adapter length after the in-place one-pass form for 17..99 bases; clip parity tests on the GPU
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r03u; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python scripts/clip_by_adapter_len.py > $O/clip_by_adapter_len.txt 2> $O/clip_by_adapter_len.err; echo "adapter rc=$?"; cat $O/clip_by_adapter_len.txt; tail -3 $O/clip_by_adapter_len.err
L=150 timeout 600 python scripts/clip_by_adapter_len.py 13 20 24 34 48 64 > $O/clip_by_adapter_len_150.txt 2>> $O/clip_by_adapter_len.err; cat $O/clip_by_adapter_len_150.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fuzz or configs or variable or long_reads or clip or adversarial" > $O/pytest_clip.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_clip.log
