#!/bin/bash
# round-2 GPU call J: fxg_kernel_rows (one lane per read, rows in registers) -- parity first, then A/B against the tile kernel
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02j; mkdir -p $O
export TMPDIR=/tmp
echo "== parity"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -8 | tee $O/pytest.txt
echo "== A/B"
for r in 1 0; do
  echo "FXG_ROWS=$r"
  FXG_ROWS=$r timeout 300 python bench.py --steps 20 --warmup 3 2>&1 | grep -v amdgpu.ids | cut -c1-700
done | tee $O/ab.txt
