#!/bin/bash
# round 6, call af: the statistics kernel, round robin by trip, with its row loads 1, 2, 3 and 4 trips ahead of the adds; first the statistics tests on the depth-2 build
O=gpurun_out/r06af; mkdir -p $O
export FXG_QS_CHUNK_TRIPS=0xFFFFFFFF
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "quality_stats or long_reads" > $O/pytest_stats.txt 2>&1; tail -n 2 $O/pytest_stats.txt
for rep in 1 2 3 4; do
for v in libfxg_v_qsd1.so libfxg.so libfxg_v_qsd3.so libfxg_v_qsd4.so; do
  echo -n "$v: "; FXG_LIB=$PWD/fastx_toolkit_amd/$v timeout 300 python scripts/bench_stats.py 2>&1 | tail -n 1 | cut -c1-110
done; done > $O/stats_rr_depth.txt 2>&1
cat $O/stats_rr_depth.txt
