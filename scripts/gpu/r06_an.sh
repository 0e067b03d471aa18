#!/bin/bash
# round 6, call an: FETCH_SIZE of the statistics kernel by trip order (FXG_QS_ROUND_ROBIN=1 round robin, 2 contiguous slices through the same loop); the read-stream
# microbenchmark with cfg4's shape
O=gpurun_out/r06an; mkdir -p $O
for k in 1 2; do
  FXG_QS_ROUND_ROBIN=$k bash scripts/gpu/r03_pmc.sh r06an/pmc_rr$k "stats" > $O/pmc_rr$k.log 2>&1
  python scripts/pmc_traffic.py gpurun_out/r06an/pmc_rr$k/stats r06an stats 2>&1 | grep -o '"traffic_over_algorithmic": [0-9.]*' | sed "s/^/order $k: /"
done | tee $O/stats_traffic_by_order.txt
git checkout profiles/pmc_traffic_stats.json 2>/dev/null
timeout 600 scripts/ubench/read_stream 2>&1 | grep "copy\|mix" | tee $O/read_stream_copy.txt
