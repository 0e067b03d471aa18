#!/bin/bash
# round-2 GPU call P: fxg_kernel_rows, two tiles per wave one step apart -- parity, then clocks
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02p; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest.txt
export ABLATE='[["rows full",{}],["tiles full",{"FXG_ROWS":"0"}],["no stores",{"FXG_DEBUG":"1"}],["rows 8/cu",{"FXG_BLOCKS_PER_CU":"8"}]]'
VARIANTS="abl abl_lb4" timeout 600 python scripts/variants.py run 2>&1 | grep -v amdgpu.ids | tee $O/ablate.txt
