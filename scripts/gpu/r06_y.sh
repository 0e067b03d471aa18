#!/bin/bash
# round 6, call y: start / end-of-loop / end clocks of every workgroup of the statistics kernel (-DFXG_QS_CLOCKS build): how evenly the static slices finish
O=gpurun_out/r06y; mkdir -p $O
FXG_LIB=$PWD/fastx_toolkit_amd/libfxg_v_qsclk.so timeout 300 python scripts/stats_wg_clocks.py > $O/stats_wg_clocks.txt 2>&1
cat $O/stats_wg_clocks.txt | cut -c1-300
