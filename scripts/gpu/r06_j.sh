#!/bin/bash
# round 6, call j: the statistics kernel with its loads 1 / 2 / 3 / 4 trips ahead of its LDS adds; its parity test
O=gpurun_out/r06j; mkdir -p $O
for v in qs1 qs2 default qs4; do lib=fastx_toolkit_amd/libfxg_v_$v.so; [ $v = default ] && lib=fastx_toolkit_amd/libfxg.so
  echo -n "$v (depth: qs1 1, qs2 2, default 3, qs4 4): "; FXG_LIB=$PWD/$lib timeout 300 python scripts/bench_stats.py 2>/dev/null | tail -1; done > $O/stats_depth.txt 2>&1
cat $O/stats_depth.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "stats" > $O/pytest_stats.txt 2>&1; tail -n 3 $O/pytest_stats.txt
