#!/bin/bash
# round-2 GPU call AD: clip kernel, the row's base loaded one row ahead -- parity, then cfg3 / cfg5 kernel times
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02ad; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fuzz or configs or variable or history or cfg3 or cfg5" 2>&1 | tail -3 | tee $O/pytest.txt
for c in cfg3 cfg5; do ONLY=$c timeout 300 python scripts/bench_configs.py 2>&1 | grep -v amdgpu.ids | cut -c1-260; done | tee $O/clip.txt
