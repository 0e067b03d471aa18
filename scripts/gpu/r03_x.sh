#!/bin/bash
# round-3 GPU call X: the whole GPU tier at HEAD
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r03x; mkdir -p $O
export TMPDIR=/tmp
free -g | head -2; df -h /tmp | tail -1
timeout 1500 python -m pytest tests -x -q -m gpu --durations=8 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -14 $O/pytest_gpu.log
free -g | head -2
