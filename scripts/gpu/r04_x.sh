#!/bin/bash
# round 4, call x: issue classes of the VALU instructions the DP is made of (scripts/ubench/valu_rate), and the shader clock the clip kernel runs at
mkdir -p gpurun_out/r04x
for i in 0 1 2 4 40 41 42 43 44 45 46 47 48 49 50 51 52 53 54 55 56 57 58 35 36 59 14 20 23; do timeout 30 scripts/ubench/valu_rate $i; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04x/valu_rate.txt
for c in cfg3 cfg5; do LIBS=fastx_toolkit_amd/libfxg_x_abl.so CFG=$c timeout 280 python scripts/ablate_clip.py 2>&1 | grep -v "amdgpu.ids\|scanner"; done | tee gpurun_out/r04x/ablate_clock.txt
