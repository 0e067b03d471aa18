#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02v; mkdir -p $O
export TMPDIR=/tmp
export ABLATE='[["rows full",{}],["no wait no stores",{"FXG_DEBUG":"3"}],["tiles full",{"FXG_ROWS":"0"}],["tiles t256",{"FXG_ROWS":"0","FXG_TILE":"256"}]]'
timeout 900 python scripts/variants.py run 2>&1 | grep -v amdgpu.ids | tee $O/ablate.txt
