#!/bin/bash
# round-2 GPU call O: dot4 masks + shared threshold -- parity, then both kernels
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02o; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest.txt
FXG_ROWS=0 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3 | tee -a $O/pytest.txt
export ABLATE='[["rows full",{}],["tiles full",{"FXG_ROWS":"0"}],["tiles t256",{"FXG_ROWS":"0","FXG_TILE":"256"}],["no stores",{"FXG_DEBUG":"1"}]]'
VARIANTS="abl" timeout 600 python scripts/variants.py run 2>&1 | grep -v amdgpu.ids | tee $O/ablate.txt
