#!/bin/bash
# round 4, call r: pair values by letter class through the VGPR index mode (pass 1 of the 4..16-column clip instances): timing and parity
mkdir -p gpurun_out/r04r
for v in ship pclass; do
  lib=fastx_toolkit_amd/libfxg_x_$v.so; [ $v = ship ] && lib=fastx_toolkit_amd/libfxg.so
  for cfg in cfg3 cfg5; do echo "$v $(FXG_LIB=$lib python scripts/clip_roles_potential.py $cfg 1 2>/dev/null)"; done
done | tee gpurun_out/r04r/clip_times.txt
FXG_LIB=$PWD/fastx_toolkit_amd/libfxg_x_pclass.so timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "fuzz or adversarial or history or variable or long_reads or configs_vs" 2>&1 | tail -8
