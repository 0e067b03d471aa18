#!/bin/bash
# round-2 GPU call N: phase clocks of fxg_kernel_rows
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02n; mkdir -p $O
export TMPDIR=/tmp
export ABLATE='[["rows full",{}],["no stores",{"FXG_DEBUG":"1"}],["no wait",{"FXG_DEBUG":"2"}],["rows 8/cu",{"FXG_BLOCKS_PER_CU":"8"}]]'
VARIANTS="abl abl_k8" timeout 600 python scripts/variants.py run 2>&1 | grep -v amdgpu.ids | tee $O/ablate.txt
