#!/bin/bash
# round 6, call x: the statistics kernel's new flush with the loads 1, 2 and 4 trips ahead, three rounds (run-to-run spread)
O=gpurun_out/r06x; mkdir -p $O
for rep in 1 2 3; do
for v in libfxg.so libfxg_v_qsd2.so libfxg_v_qsd4.so libfxg_v_qs1.so; do
  echo -n "$v: "; FXG_LIB=$PWD/fastx_toolkit_amd/$v timeout 300 python scripts/bench_stats.py 2>&1 | tail -n 1
done; done > $O/stats_depth_new_flush.txt 2>&1
cat $O/stats_depth_new_flush.txt | cut -c1-120
