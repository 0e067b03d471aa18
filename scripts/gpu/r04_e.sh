#!/bin/bash
mkdir -p gpurun_out/r04e
for v in m_w4 m_w3; do
  FXG_LIB=fastx_toolkit_amd/libfxg_$v.so timeout 600 python scripts/debug/clip_wide_seq.py > gpurun_out/r04e/seq_$v.txt 2>&1
  FXG_LIB=fastx_toolkit_amd/libfxg_$v.so SKIP_ADVERSARIAL=1 timeout 600 python scripts/debug/clip_wide_seq.py > gpurun_out/r04e/seq_noadv_$v.txt 2>&1
done
grep -v " ok$" gpurun_out/r04e/seq_m_w4.txt | head -40; echo ===; grep -v " ok$" gpurun_out/r04e/seq_noadv_m_w4.txt | head -30; echo === w3; grep -v " ok$" gpurun_out/r04e/seq_m_w3.txt | head -30
