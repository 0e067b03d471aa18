#!/bin/bash
# round 4, call v: is the pass-1 row bound by instruction bytes?  asm rows with 36 and with 24 bytes per cell against the compiler's
mkdir -p gpurun_out/r04v
for v in ship asm_row asm_row2; do
  lib=fastx_toolkit_amd/libfxg_x_$v.so; [ $v = ship ] && lib=fastx_toolkit_amd/libfxg.so
  for cfg in cfg3 cfg5; do echo "$v $(FXG_LIB=$lib python scripts/clip_roles_potential.py $cfg 1 2>&1 | tail -1 | cut -c1-150)"; done
done | tee gpurun_out/r04v/clip_times.txt
FXG_LIB=$PWD/fastx_toolkit_amd/libfxg_x_asm_row2.so timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "fuzz or adversarial or history or variable or long_reads or configs_vs" 2>&1 | tail -3
