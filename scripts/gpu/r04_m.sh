#!/bin/bash
mkdir -p gpurun_out/r04m
for d in 3 4; do for cfg in cfg3 cfg5; do FXG_CLIP_DEPTH_RT=$d python scripts/clip_roles_potential.py $cfg 1 2>/dev/null | sed "s/^/depth $d /"; done; done | tee gpurun_out/r04m/depth.txt
FXG_CLIP_DEPTH_RT=4 timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "fuzz or adversarial or cfg3 or cfg5 or configs_vs or rows_kernel_keeps" 2>&1 | tail -4
