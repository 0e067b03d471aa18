#!/bin/bash
# round 4, call ae: the clip DP straight over the batch in global memory (clip_global) against the staged tile, per read length; parity of both
mkdir -p gpurun_out/r04ae
python scripts/debug/clip_global_vs_staged.py 2>&1 | grep -v amdgpu | tee gpurun_out/r04ae/clip_global_vs_staged.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "over_the_batch or fuzz or adversarial or history or variable or long_reads or configs_vs or first_n" 2>&1 | tail -3
