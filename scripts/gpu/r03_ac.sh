#!/bin/bash
# round-3 GPU call AC: the whole GPU tier in one pytest run (as the driver runs it), smoke(), then the closing evidence (r03_z.sh)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r03ac; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -x -q -m gpu --durations=8 > $O/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -14 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
[ $rc -eq 0 ] && bash scripts/gpu/r03_z.sh
