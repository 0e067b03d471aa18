#!/bin/bash
mkdir -p gpurun_out/r04h
for i in 1 2 3 4; do
FXG_LIB=fastx_toolkit_amd/libfxg_m_w4.so SKIP_ADVERSARIAL=1 ONLY=-100,-348 timeout 600 python scripts/debug/clip_wide_seq.py > gpurun_out/r04h/seq_plain_$i.txt 2>&1
FXG_LIB=fastx_toolkit_amd/libfxg_m_w4.so SKIP_ADVERSARIAL=1 ONLY=-100,-348 FRESH=1 timeout 600 python scripts/debug/clip_wide_seq.py > gpurun_out/r04h/seq_fresh_$i.txt 2>&1
FXG_LIB=fastx_toolkit_amd/libfxg_m_w3.so SKIP_ADVERSARIAL=1 ONLY=-100,-348 timeout 600 python scripts/debug/clip_wide_seq.py > gpurun_out/r04h/seq_w3_plain_$i.txt 2>&1
FXG_LIB=fastx_toolkit_amd/libfxg.so SKIP_ADVERSARIAL=1 ONLY=-100,-348 timeout 600 python scripts/debug/clip_wide_seq.py > gpurun_out/r04h/seq_ship_plain_$i.txt 2>&1
done
cd gpurun_out/r04h; grep -H "348" *.txt | grep -v " ok$" | cut -c1-220
