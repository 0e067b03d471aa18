#!/bin/bash
# round 6, call aa: the statistics kernel's row loads with the non-temporal policy (libfxg_v_qsntl.so) against the default policy (libfxg.so), eight rounds
O=gpurun_out/r06aa; mkdir -p $O
for rep in 1 2 3 4 5 6 7 8; do
for v in libfxg.so libfxg_v_qsntl.so; do
  echo -n "$v: "; FXG_LIB=$PWD/fastx_toolkit_amd/$v timeout 300 python scripts/bench_stats.py 2>&1 | tail -n 1
done; done > $O/stats_nt_loads.txt 2>&1
cut -c1-120 $O/stats_nt_loads.txt
