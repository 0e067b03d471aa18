#!/bin/bash
# round-2 GPU call L: scanner with DPP scans -- parity, then the ablations again
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02l; mkdir -p $O
export TMPDIR=/tmp
echo "== parity"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -8 | tee $O/pytest.txt
FXG_ROWS=0 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3 | tee -a $O/pytest.txt
export ABLATE='[["rows full",{}],["tiles full",{"FXG_ROWS":"0"}],["tiles full t256",{"FXG_ROWS":"0","FXG_TILE":"256"}],["tiles full t64",{"FXG_ROWS":"0","FXG_TILE":"64"}],["no stores",{"FXG_DEBUG":"1"}],["no wait",{"FXG_DEBUG":"2"}],["no stores no wait",{"FXG_DEBUG":"3"}],["no bases",{"FXG_DEBUG":"4"}],["aligned stores",{"FXG_DEBUG":"32"}],["decision only",{},false]]'
VARIANTS=abl timeout 600 python scripts/variants.py run 2>&1 | grep -v amdgpu.ids | tee $O/ablate.txt
export ABLATE='[["rows full",{}],["no stores",{"FXG_DEBUG":"1"}]]'
VARIANTS="abl_k8 abl_k32" timeout 300 python scripts/variants.py run 2>&1 | grep -v amdgpu.ids | tee -a $O/ablate.txt
