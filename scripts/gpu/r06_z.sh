#!/bin/bash
# round 6, call z: the statistics kernel with its reads dealt out in chunks: the statistics tests of the GPU tier (sizes from 1 read up), timing against the round's
# earlier builds, and the workgroups' end clocks
O=gpurun_out/r06z; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli.py -q -m gpu -k "stats" > $O/pytest_stats.txt 2>&1; tail -n 3 $O/pytest_stats.txt
for rep in 1 2 3; do
for v in libfxg.so libfxg_v_qsnoflush.so libfxg_v_qs1.so; do
  echo -n "$v: "; FXG_LIB=$PWD/fastx_toolkit_amd/$v timeout 300 python scripts/bench_stats.py 2>&1 | tail -n 1
done; done > $O/stats_dealt.txt 2>&1
cut -c1-120 $O/stats_dealt.txt
FXG_LIB=$PWD/fastx_toolkit_amd/libfxg_v_qsclk.so timeout 300 python scripts/stats_wg_clocks.py > $O/stats_wg_clocks.txt 2>&1
cat $O/stats_wg_clocks.txt | cut -c1-300
