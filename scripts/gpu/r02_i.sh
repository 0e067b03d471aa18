#!/bin/bash
# round-2 GPU call I: clip occupancy / tile A/B, new CLI tests, end-to-end after staggered lane start
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02i; mkdir -p $O
export TMPDIR=/tmp
make -s -C fastx_toolkit_amd/host 2>/dev/null
cat /sys/kernel/mm/transparent_hugepage/shmem_enabled /sys/kernel/mm/transparent_hugepage/enabled 2>/dev/null | tee $O/thp.txt
echo "== clip A/B"
for v in base clipw5 clipw3; do for t in "" 128; do
  echo "variant=$v tile=${t:-default}"
  FXG_LIB=fastx_toolkit_amd/libfxg_v_$v.so FXG_TILE=$t ONLY=cfg3 timeout 200 python scripts/bench_configs.py 2>&1 | grep -v amdgpu.ids | cut -c1-260
  FXG_LIB=fastx_toolkit_amd/libfxg_v_$v.so FXG_TILE=$t ONLY=cfg5 timeout 200 python scripts/bench_configs.py 2>&1 | grep -v amdgpu.ids | cut -c1-260
done; done | tee $O/clip.txt
echo "== new CLI tests"
timeout 600 python -m pytest tests/test_gpu_cli.py -m gpu -q -k "longer or lanes or flag" 2>&1 | tail -5 | tee $O/pytest.txt
echo "== e2e"
timeout 600 python scripts/e2e_breakdown.py 16000000 2>&1 | grep -v amdgpu.ids | grep -E "tiny|lanes=1 -> tmpfs|lanes=2 -> tmpfs file|lanes=2 -> /dev/null|fused" | tee $O/e2e.txt
