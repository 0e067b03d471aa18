#!/bin/bash
# round 6, call av: the statistics kernel's piece form with non-temporal row loads (-DFXG_QS_NTL: its aligned loads share no line between waves, so the policy should
# no longer cost fetched bytes) and with its loads 2 / 4 trips ahead, against HEAD (default policy, 3 trips); time and FETCH_SIZE
O=gpurun_out/r06av; mkdir -p $O
P=$PWD/fastx_toolkit_amd
for rep in 1 2 3 4; do for v in libfxg.so libfxg_v_qsntl.so libfxg_v_qsd2.so libfxg_v_qsd4.so libfxg_v_qsntld4.so; do
  echo -n "$v: "; FXG_LIB=$P/$v timeout 300 python scripts/bench_stats.py 2>&1 | tail -n 1 | cut -c1-110
done; done | tee $O/stats_piece_nt_depth.txt
for v in libfxg_v_qsntl.so; do
  FXG_LIB=$P/$v bash scripts/gpu/r03_pmc.sh r06av/pmc_$v "stats" > $O/pmc_$v.log 2>&1
  python scripts/pmc_traffic.py gpurun_out/r06av/pmc_$v/stats r06av stats 2>&1 | grep -o '"traffic_over_algorithmic": [0-9.]*' | sed "s/^/$v: /"
done | tee $O/stats_piece_nt_traffic.txt
git checkout profiles/pmc_traffic_stats.json 2>/dev/null
