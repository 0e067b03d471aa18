#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02t; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k both_quality 2>&1 | tail -5 | tee $O/rvt.txt   # was scripts/rows_vs_tiles.py, now this test
export ABLATE='[["rows full",{}],["tiles full",{"FXG_ROWS":"0"}],["no stores",{"FXG_DEBUG":"1"}]]'
VARIANTS="abl abl_lb4" timeout 600 python scripts/variants.py run 2>&1 | grep -v amdgpu.ids | tee $O/ablate.txt
