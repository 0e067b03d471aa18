#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02y; mkdir -p $O
export TMPDIR=/tmp
export ABLATE='[["rows full",{}],["tiles full",{"FXG_ROWS":"0"}],["rows full",{}],["no stores",{"FXG_DEBUG":"1"}]]'
timeout 900 python scripts/variants.py run 2>&1 | grep -v amdgpu.ids | tee $O/ablate.txt
