#!/bin/bash
# round-3 GPU call I: clip kernel after mask hoisting / 3 slots / shared bitmap / gather K=4: parity subset, phase clocks, kernel times
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
O=$R/gpurun_out/r03i; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fuzz or configs_vs or variable_length or clip_history or cfg3 or cfg5 or long_reads" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for c in cfg3 cfg5; do LIBS=build/libfxg_abl256_d3.so,build/libfxg_abl256_d3_gk1.so,build/libfxg_abl64_d3.so CFG=$c timeout 280 python scripts/ablate_clip.py 2>&1 | grep -v "amdgpu.ids\|scanner"; done | tee $O/ablate.txt
for c in cfg3 cfg5; do ONLY=$c timeout 300 python scripts/bench_configs.py 2>&1 | grep -v amdgpu.ids | tee -a $O/times.txt | cut -c1-330; done
