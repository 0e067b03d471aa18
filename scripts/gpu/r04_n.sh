#!/bin/bash
# round 4, call n: wave-uniform DP loops -- matrix (every instance at 2, 3, 4 waves), clip timings, clip parity subset
mkdir -p gpurun_out/r04n
for cfg in cfg3 cfg5; do python scripts/clip_roles_potential.py $cfg 1 2>/dev/null; python scripts/clip_roles_potential.py $cfg 0 2>/dev/null; done | tee gpurun_out/r04n/clip_times.txt
timeout 2400 python -m pytest tests/test_gpu_clip_matrix.py -q -s -m gpu > gpurun_out/r04n/pytest_matrix.txt 2>&1; grep -a "^waves\|passed\|failed" gpurun_out/r04n/pytest_matrix.txt | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "fuzz or adversarial or history or variable or long_reads or configs_vs" 2>&1 | tail -3
