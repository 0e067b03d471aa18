#!/bin/bash
# round 6, call b: issue rate of v_fma_mix_f32 and of the table cells; the clip kernel with its pair table as halves (v_fma_mix_f32) / floats (v_add_f32), with and without a scheduling fence
O=gpurun_out/r06b; mkdir -p $O
for i in 0 2 77 78 79 80 81 82 17; do timeout 60 scripts/ubench/valu_rate $i; done > $O/valu_rate_fma_mix.txt 2>&1
cat $O/valu_rate_fma_mix.txt
LIBS=fastx_toolkit_amd/libfxg_v_noptab.so,fastx_toolkit_amd/libfxg.so,fastx_toolkit_amd/libfxg_v_sched.so,fastx_toolkit_amd/libfxg_v_f32.so,fastx_toolkit_amd/libfxg_v_f32s.so timeout 900 python scripts/clip_ab.py > $O/clip_ab.txt 2>&1
cut -c1-400 $O/clip_ab.txt
