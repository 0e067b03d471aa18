#!/bin/bash
# round-3 GPU call AB: the register-form clipper on reads of any length (clip parity tests, 250/300-base reads), then the closing evidence (r03_z.sh)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r03ab; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fuzz or configs or variable or long_reads or clip or adversarial" > $O/pytest_clip.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest_clip.log
L=300 READS=10000000 timeout 600 python scripts/clip_by_adapter_len.py 8 13 16 20 > $O/clip_by_adapter_len_300.txt 2> $O/err.txt; cut -c1-150 $O/clip_by_adapter_len_300.txt
L=1000 READS=2000000 timeout 600 python scripts/clip_by_adapter_len.py 13 34 > $O/clip_by_adapter_len_1000.txt 2>> $O/err.txt; cut -c1-150 $O/clip_by_adapter_len_1000.txt
[ $rc -eq 0 ] && bash scripts/gpu/r03_z.sh
