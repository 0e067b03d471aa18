#!/bin/bash
# round-3 GPU call B: in-place score rows / scalar loops / untracked pass-2 rows: parity subset, kernel times, SQ counters of the clip kernel
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
O=$R/gpurun_out/r03b; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fuzz or configs_vs or variable_length or clip_history or cfg3 or cfg5 or long_reads" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for c in cfg3 cfg5; do ONLY=$c timeout 300 python scripts/bench_configs.py 2>&1 | grep -v amdgpu.ids | tee -a $O/times.txt | cut -c1-330; done
bash scripts/pmc_sq.sh r03b/pmc_clip scripts/pmc_clip.py
