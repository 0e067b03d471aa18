#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02w; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k both_quality 2>&1 | tail -5 | tee $O/rvt.txt   # was scripts/rows_vs_tiles.py, now this test
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest.txt
export ABLATE='[["rows full",{}],["no stores",{"FXG_DEBUG":"1"}],["tiles full",{"FXG_ROWS":"0"}]]'
timeout 900 python scripts/variants.py run 2>&1 | grep -v amdgpu.ids | tee $O/ablate.txt
