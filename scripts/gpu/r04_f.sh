#!/bin/bash
mkdir -p gpurun_out/r04f
for i in 1 2 3; do
FXG_LIB=fastx_toolkit_amd/libfxg_m_w4.so SKIP_ADVERSARIAL=1 ONLY=-348 SCRAMBLE=1 timeout 600 python scripts/debug/clip_wide_seq.py > gpurun_out/r04f/seq348_scr_$i.txt 2>&1
FXG_LIB=fastx_toolkit_amd/libfxg_m_w4.so SKIP_ADVERSARIAL=1 ONLY=-348 timeout 600 python scripts/debug/clip_wide_seq.py > gpurun_out/r04f/seq348_plain_$i.txt 2>&1
done
FXG_LIB=fastx_toolkit_amd/libfxg_m_w3.so SKIP_ADVERSARIAL=1 SCRAMBLE=1 timeout 600 python scripts/debug/clip_wide_seq.py > gpurun_out/r04f/seq_w3_scr.txt 2>&1
FXG_LIB=fastx_toolkit_amd/libfxg_m_w2.so SKIP_ADVERSARIAL=1 SCRAMBLE=1 timeout 600 python scripts/debug/clip_wide_seq.py > gpurun_out/r04f/seq_w2_scr.txt 2>&1
FXG_LIB=fastx_toolkit_amd/libfxg.so SKIP_ADVERSARIAL=1 SCRAMBLE=1 timeout 600 python scripts/debug/clip_wide_seq.py > gpurun_out/r04f/seq_ship_scr.txt 2>&1
cd gpurun_out/r04f; grep -H -v "^    read\| ok$\|amdgpu.ids\|adversarial cases" *.txt | cut -c1-220
