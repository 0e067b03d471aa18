#!/bin/bash
# round 6, call ao: FETCH_SIZE and time of the statistics kernel with its loads 1, 2 and 3 trips ahead (what the deeper prefetch costs in fetched bytes)
O=gpurun_out/r06ao; mkdir -p $O
for v in libfxg_v_qsd1.so libfxg_v_qsd2.so libfxg.so; do
  FXG_LIB=$PWD/fastx_toolkit_amd/$v bash scripts/gpu/r03_pmc.sh r06ao/pmc_$v "stats" > $O/pmc_$v.log 2>&1
  python scripts/pmc_traffic.py gpurun_out/r06ao/pmc_$v/stats r06ao stats 2>&1 | grep -o '"traffic_over_algorithmic": [0-9.]*' | sed "s/^/$v: /"
done | tee $O/stats_traffic_by_depth.txt
git checkout profiles/pmc_traffic_stats.json 2>/dev/null
for rep in 1 2 3 4; do for v in libfxg_v_qsd1.so libfxg_v_qsd2.so libfxg.so; do
  echo -n "$v: "; FXG_LIB=$PWD/fastx_toolkit_amd/$v timeout 300 python scripts/bench_stats.py 2>&1 | tail -n 1 | cut -c1-110
done; done | tee $O/stats_time_by_depth.txt
