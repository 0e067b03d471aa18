#!/bin/bash
# round-2 GPU call A: microbenchmark, parity tests, A/B of the cfg2 kernel variants, per-config kernel times, bench line
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02a; mkdir -p $O
export TMPDIR=/tmp
( timeout 120 scripts/ubench/valu_rate > $O/valu_rate.txt 2>&1 ) 
echo "== valu_rate"; cat $O/valu_rate.txt | tail -25
echo "== gpu tests"
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 | tee $O/pytest.txt
echo "== variants"
TWO='[["qlds0 t256",{"FXG_QLDS":"0"}],["qlds1 t128",{"FXG_QLDS":"1","FXG_QLDS_BUDGET":"53248"}],["qlds1 t64",{"FXG_QLDS":"1","FXG_QLDS_BUDGET":"26624"}],["qlds1 t32",{"FXG_QLDS":"1","FXG_QLDS_BUDGET":"13312"}],["qlds1 t256",{"FXG_QLDS":"1","FXG_QLDS_BUDGET":"104448"}],["qlds0 t128",{"FXG_QLDS":"0","FXG_TILE":"128"}],["decision-only",{},false]]'
ONE='[["qlds0 t256",{"FXG_QLDS":"0"}],["qlds1 t128",{"FXG_QLDS":"1","FXG_QLDS_BUDGET":"28672"}],["qlds1 t256",{"FXG_QLDS":"1","FXG_QLDS_BUDGET":"57344"}],["qlds1 t64",{"FXG_QLDS":"1","FXG_QLDS_BUDGET":"15360"}],["qlds0 t128",{"FXG_QLDS":"0","FXG_TILE":"128"}]]'
( VARIANTS="scan lookback scan_k2w3 scan_u8w3" ABLATE="$TWO" timeout 600 python scripts/variants.py run; VARIANTS="same same_k2w4" ABLATE="$ONE" timeout 300 python scripts/variants.py run ) 2>&1 | grep -v amdgpu.ids | tee $O/variants.txt
echo "== configs"
timeout 600 python scripts/bench_configs.py 2>&1 | grep -v amdgpu.ids | tee $O/configs.txt
timeout 300 python scripts/bench_stats.py 2>&1 | grep -v amdgpu.ids | tee $O/stats.txt
echo "== bench"
timeout 600 python bench.py --steps 20 --warmup 3 2>&1 | grep -v amdgpu.ids | tee $O/bench.json
