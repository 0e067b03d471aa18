#!/bin/bash
# round 6, call w: the statistics kernel with the store / fire-and-forget flush (libfxg.so) against no flush traffic at all (libfxg_v_qsnoflush.so, wrong counts),
# the round's earlier build (libfxg_v_qs1.so: `+=` flush) and the loads alone (libfxg_v_qsnoacc1.so); then the statistics test of the GPU tier
O=gpurun_out/r06w; mkdir -p $O
for rep in 1 2; do
for v in libfxg.so libfxg_v_qsnoflush.so libfxg_v_qs1.so libfxg_v_qsnoacc1.so; do
  echo -n "$v: "; FXG_LIB=$PWD/fastx_toolkit_amd/$v timeout 300 python scripts/bench_stats.py 2>&1 | tail -n 1
done; done > $O/stats_flush.txt 2>&1
cat $O/stats_flush.txt | cut -c1-250
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "quality_stats" > $O/pytest_stats.txt 2>&1; tail -n 3 $O/pytest_stats.txt
