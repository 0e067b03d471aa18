#!/bin/bash
# round 4, call z: wave maximum by DPP in the register two-pass form, best row without the tracking form: timing and parity
mkdir -p gpurun_out/r04z
for cfg in cfg3 cfg5; do python scripts/clip_roles_potential.py $cfg 1 2>&1 | tail -1 | cut -c1-150; done | tee gpurun_out/r04z/clip_times.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "fuzz or adversarial or history or variable or long_reads or configs_vs" 2>&1 | tail -3
