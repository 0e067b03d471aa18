#!/bin/bash
# round-3 GPU call A: two-pass clipper -- parity (fuzz, configs, cfg3/cfg5 full size) and kernel time against the one-pass build
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
O=$R/gpurun_out/r03a; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fuzz or configs_vs or variable_length or clip_history or cfg3 or cfg5 or long_reads" --durations=8 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log
for lib in "" build/libfxg_onepass.so; do
  echo "== lib=${lib:-default}"
  FXG_LIB=${lib:+$R/$lib} ONLY=cfg3 timeout 300 python scripts/bench_configs.py 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_${lib:+onepass}.txt | cut -c1-400
  FXG_LIB=${lib:+$R/$lib} ONLY=cfg5 timeout 300 python scripts/bench_configs.py 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_${lib:+onepass}.txt | cut -c1-400
done
