#!/bin/bash
# round-3 GPU call Y: one file of the GPU tier per call (FILES="tests/test_gpu_cli.py" ...), with the box's memory before and after
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r03y; mkdir -p $O
export TMPDIR=/tmp
free -g | sed -n 1,2p; df -h /tmp /dev/shm | tail -2
timeout ${LIMIT:-900} python -m pytest ${FILES:-tests/test_gpu_cli.py} -x -q -m gpu --durations=8 ${EXTRA:-} > $O/pytest_${TAG:-cli}.log 2>&1; echo "pytest rc=$?"; tail -14 $O/pytest_${TAG:-cli}.log
free -g | sed -n 1,2p
