#!/bin/bash
# round-3 GPU call AA: adapters that contain N through the packed forms (instances -3xx): parity, then times against the general form
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r03aa; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fuzz or configs or variable or long_reads or clip or adversarial" > $O/pytest_clip.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_clip.log
WITH_N=1 timeout 600 python scripts/clip_by_adapter_len.py 13 24 34 48 64 > $O/clip_with_n_100.txt 2> $O/err.txt; cut -c1-170 $O/clip_with_n_100.txt
WITH_N=1 L=150 timeout 600 python scripts/clip_by_adapter_len.py 13 34 64 > $O/clip_with_n_150.txt 2>> $O/err.txt; cut -c1-170 $O/clip_with_n_150.txt
WITH_N=1 FXG_NO_PACKED_CLIP=1 timeout 600 python scripts/clip_by_adapter_len.py 13 34 64 > $O/clip_with_n_100_general.txt 2>> $O/err.txt; cut -c1-170 $O/clip_with_n_100_general.txt
tail -2 $O/err.txt
