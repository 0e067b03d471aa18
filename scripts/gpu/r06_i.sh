#!/bin/bash
# round 6, call i: shader cycles per row of the DP, as its waves see them
O=gpurun_out/r06i; mkdir -p $O
FXG_LIB=$PWD/fastx_toolkit_amd/libfxg_v_rowclk.so timeout 600 python scripts/clip_rowclk.py > $O/rowclk.txt 2>&1
FXG_DEBUG=48 FXG_LIB=$PWD/fastx_toolkit_amd/libfxg_v_rowclk.so timeout 600 python scripts/clip_rowclk.py >> $O/rowclk.txt 2>&1
cut -c1-420 $O/rowclk.txt
