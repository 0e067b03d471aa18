#!/bin/bash
# round-2 GPU call K: where does fxg_kernel_rows spend its time (FXG_DEBUG ablations; results are wrong when a bit is set)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02k; mkdir -p $O
export TMPDIR=/tmp
export ABLATE='[["rows full",{}],["tiles full",{"FXG_ROWS":"0"}],["no stores",{"FXG_DEBUG":"1"}],["no wait",{"FXG_DEBUG":"2"}],["no stores no wait",{"FXG_DEBUG":"3"}],["no bases",{"FXG_DEBUG":"4"}],["aligned stores",{"FXG_DEBUG":"32"}],["aligned stores no wait",{"FXG_DEBUG":"34"}],["decision only",{},false]]'
VARIANTS=abl timeout 600 python scripts/variants.py run 2>&1 | grep -v amdgpu.ids | tee $O/ablate.txt
export ABLATE='[["rows full",{}],["no wait",{"FXG_DEBUG":"2"}]]'
VARIANTS="abl_k8 abl_k32" timeout 300 python scripts/variants.py run 2>&1 | grep -v amdgpu.ids | tee -a $O/ablate.txt
