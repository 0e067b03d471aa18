#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02u; mkdir -p $O
export TMPDIR=/tmp
export ABLATE='[["rows full",{}],["no wait no stores",{"FXG_DEBUG":"3"}],["rows 8/cu",{"FXG_BLOCKS_PER_CU":"8"}],["rows 4/cu",{"FXG_BLOCKS_PER_CU":"4"}]]'
timeout 900 python scripts/variants.py run 2>&1 | grep -v amdgpu.ids | tee $O/ablate.txt
