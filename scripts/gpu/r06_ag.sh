#!/bin/bash
# round 6, call ag: cfg2's row loads (LDS-DMA) with the non-temporal policy against the default, re-measured at HEAD (round 2 found no gain); then the statistics tests
# of the GPU tier on the round-robin build
O=gpurun_out/r06ag; mkdir -p $O
for rep in 1 2 3 4; do
for v in libfxg.so libfxg_v_rowsnt.so; do
  echo -n "$v: "; ONLY=cfg2 FXG_LIB=$PWD/fastx_toolkit_amd/$v timeout 300 python scripts/bench_configs.py 2>&1 | tail -n 1 | cut -c1-200
done; done > $O/cfg2_rows_nt.txt 2>&1
cat $O/cfg2_rows_nt.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli.py -q -m gpu -k "stats or long_reads" > $O/pytest_stats.txt 2>&1; tail -n 2 $O/pytest_stats.txt
for rep in 1 2 3; do FXG_LIB=$PWD/fastx_toolkit_amd/libfxg.so timeout 300 python scripts/bench_stats.py 2>&1 | tail -n 1 | cut -c1-110; done | tee $O/stats_head.txt
