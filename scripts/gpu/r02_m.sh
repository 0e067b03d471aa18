#!/bin/bash
# round-2 GPU call M: fxg_kernel_rows with packed write-out through LDS -- parity, then ablations
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02m; mkdir -p $O
export TMPDIR=/tmp
echo "== parity"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -8 | tee $O/pytest.txt
export ABLATE='[["rows full",{}],["tiles full",{"FXG_ROWS":"0"}],["no stores",{"FXG_DEBUG":"1"}],["no wait",{"FXG_DEBUG":"2"}],["no stores no wait",{"FXG_DEBUG":"3"}],["no bases",{"FXG_DEBUG":"4"}],["rows 12/cu",{"FXG_BLOCKS_PER_CU":"12"}],["rows 8/cu",{"FXG_BLOCKS_PER_CU":"8"}]]'
VARIANTS=abl timeout 600 python scripts/variants.py run 2>&1 | grep -v amdgpu.ids | tee $O/ablate.txt
export ABLATE='[["rows full",{}]]'
VARIANTS="abl_k8 abl_k32" timeout 300 python scripts/variants.py run 2>&1 | grep -v amdgpu.ids | tee -a $O/ablate.txt
