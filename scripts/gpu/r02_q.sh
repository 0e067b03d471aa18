#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02q; mkdir -p $O
export TMPDIR=/tmp
export ABLATE='[["rows full",{}],["no stores",{"FXG_DEBUG":"1"}],["rows 10/cu",{"FXG_BLOCKS_PER_CU":"10"}]]'
VARIANTS="abl" timeout 600 python scripts/variants.py run 2>&1 | grep -v amdgpu.ids | tee $O/ablate.txt
