#!/bin/bash
# round-2 GPU call C: per-test microbenchmarks, rows-in-LDS (qualities + bases) layouts, parity with them on
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02c; mkdir -p $O
export TMPDIR=/tmp
echo "== valu_rate"
for i in 3 14 20 21 22 23 24 25 26 27 28 29 30 15 16 17 18; do timeout 20 scripts/ubench/valu_rate $i 2>&1 | cut -c1-220; done | tee $O/valu_rate.txt
echo "== gpu tests (subset, FXG_QLDS=2)"
FXG_QLDS=2 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fuzz or configs or cfg2 or golden or invalid" 2>&1 | tail -6 | tee $O/pytest.txt
echo "== variants"
TWO='[["qlds0 t128",{"FXG_QLDS":"0","FXG_TILE":"128"}],["qlds0 t64",{"FXG_QLDS":"0","FXG_TILE":"64"}],["qlds0 t32",{"FXG_QLDS":"0","FXG_TILE":"32"}],["rows2 t64",{"FXG_QLDS":"2","FXG_QLDS_BUDGET":"47104"}],["rows2 t32",{"FXG_QLDS":"2","FXG_QLDS_BUDGET":"23552"}],["rows1 t64",{"FXG_QLDS":"1","FXG_QLDS_BUDGET":"26624"}]]'
ONE='[["qlds0 t256",{"FXG_QLDS":"0"}],["rows2 t64",{"FXG_QLDS":"2","FXG_QLDS_BUDGET":"24576"}],["rows2 t128",{"FXG_QLDS":"2","FXG_QLDS_BUDGET":"49152"}],["rows2 t32",{"FXG_QLDS":"2","FXG_QLDS_BUDGET":"12800"}],["rows1 t128",{"FXG_QLDS":"1","FXG_QLDS_BUDGET":"28672"}]]'
( VARIANTS="scan scan_k2w3" ABLATE="$TWO" timeout 600 python scripts/variants.py run; VARIANTS="same same_k2w4" ABLATE="$ONE" timeout 300 python scripts/variants.py run ) 2>&1 | grep -v amdgpu.ids | tee $O/variants.txt
