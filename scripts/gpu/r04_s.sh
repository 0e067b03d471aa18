#!/bin/bash
# round 4, call s: pass-1 row of the 9..16-column clip instances as one asm per column (order of ubench test 35): timing and parity
mkdir -p gpurun_out/r04s
for v in ship asm_row; do
  lib=fastx_toolkit_amd/libfxg_x_$v.so; [ $v = ship ] && lib=fastx_toolkit_amd/libfxg.so
  for cfg in cfg3 cfg5; do echo "$v $(FXG_LIB=$lib python scripts/clip_roles_potential.py $cfg 1 2>&1 | tail -1)"; done
done | tee gpurun_out/r04s/clip_times.txt
FXG_LIB=$PWD/fastx_toolkit_amd/libfxg_x_asm_row.so timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "fuzz or adversarial or history or variable or long_reads or configs_vs" 2>&1 | tail -5
