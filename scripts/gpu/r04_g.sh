#!/bin/bash
mkdir -p gpurun_out/r04g
for i in 1 2 3; do
FXG_LIB=fastx_toolkit_amd/libfxg_m_w4.so SKIP_ADVERSARIAL=1 timeout 600 python scripts/debug/clip_wide_seq.py > gpurun_out/r04g/seq_plain_$i.txt 2>&1
FXG_LIB=fastx_toolkit_amd/libfxg_m_w4.so SKIP_ADVERSARIAL=1 SCRAMBLE=1 timeout 600 python scripts/debug/clip_wide_seq.py > gpurun_out/r04g/seq_scr_$i.txt 2>&1
FXG_LIB=fastx_toolkit_amd/libfxg_m_w4.so SKIP_ADVERSARIAL=1 ONLY=-100,-348 timeout 600 python scripts/debug/clip_wide_seq.py > gpurun_out/r04g/seq_100_348_$i.txt 2>&1
done
cd gpurun_out/r04g; grep -H "348\|-48\|-32" *.txt | grep -v " ok$" | cut -c1-220
