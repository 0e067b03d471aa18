#!/bin/bash
# round-2 GPU call B: mask/LDS microbenchmarks, parity of the aligned LDS window, A/B of the quality-rows-in-LDS layouts, clip timing
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02b; mkdir -p $O
export TMPDIR=/tmp
( timeout 120 scripts/ubench/valu_rate > $O/valu_rate.txt 2>&1 )
echo "== valu_rate"; cat $O/valu_rate.txt | cut -c1-200 | tail -32
echo "== gpu tests (subset)"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fuzz or configs or cfg2 or golden or invalid" 2>&1 | tail -6 | tee $O/pytest.txt
echo "== variants"
TWO='[["qlds0 t256",{"FXG_QLDS":"0"}],["qlds0 t128",{"FXG_QLDS":"0","FXG_TILE":"128"}],["qlds1 t128",{"FXG_QLDS":"1","FXG_QLDS_BUDGET":"53248"}],["qlds1 t64",{"FXG_QLDS":"1","FXG_QLDS_BUDGET":"26624"}],["qlds1 t32",{"FXG_QLDS":"1","FXG_QLDS_BUDGET":"13312"}],["qlds1 t128 bpc2",{"FXG_QLDS":"1","FXG_QLDS_BUDGET":"53248","FXG_BLOCKS_PER_CU":"2"}]]'
ONE='[["qlds0 t256",{"FXG_QLDS":"0"}],["qlds1 t128",{"FXG_QLDS":"1","FXG_QLDS_BUDGET":"28672"}],["qlds1 t256",{"FXG_QLDS":"1","FXG_QLDS_BUDGET":"57344"}],["qlds1 t64",{"FXG_QLDS":"1","FXG_QLDS_BUDGET":"15360"}],["qlds0 t128",{"FXG_QLDS":"0","FXG_TILE":"128"}]]'
( VARIANTS="scan scan_k2w3 scan_k2w4" ABLATE="$TWO" timeout 600 python scripts/variants.py run; VARIANTS="same same_k2w4" ABLATE="$ONE" timeout 300 python scripts/variants.py run ) 2>&1 | grep -v amdgpu.ids | tee $O/variants.txt
echo "== configs"
ONLY=cfg3 timeout 300 python scripts/bench_configs.py 2>&1 | grep -v amdgpu.ids | tee $O/configs.txt
FXH_TIMING=1 timeout 300 bash scripts/cli_timing.sh 2>&1 | tail -12 | tee $O/cli.txt
