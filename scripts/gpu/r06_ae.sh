#!/bin/bash
# round 6, call ae: the statistics kernel by the order its trips are dealt in -- static slices (0), tickets for chunks of 32 / 4 / 2 trips, round robin by trip --
# and the same for its loads alone (-DFXG_QS_NOACC), three rounds
O=gpurun_out/r06ae; mkdir -p $O
for rep in 1 2 3; do
for v in libfxg.so libfxg_v_qsnoacc.so; do
for k in default 0 4 2 0xFFFFFFFF; do
  if [ $k = default ]; then unset FXG_QS_CHUNK_TRIPS; else export FXG_QS_CHUNK_TRIPS=$k; fi
  echo -n "$v trips=$k: "; FXG_LIB=$PWD/fastx_toolkit_amd/$v timeout 300 python scripts/bench_stats.py 2>&1 | tail -n 1 | cut -c1-110
done; done; done > $O/stats_order.txt 2>&1
cat $O/stats_order.txt
