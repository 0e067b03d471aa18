#!/bin/bash
# round 5, call u: the CLI campaign over the REAL engine again (the test transport ahead of the system's RCCL on the library path)
O=gpurun_out/r05u; mkdir -p $O
for s in 611 612 613; do FXG_CAMPAIGN_REAL=1 timeout 400 python scripts/fuzz_campaign_cli.py $s 170 > $O/fuzz_real_$s.txt 2>&1 & done
wait
for s in 611 612 613; do tail -n 2 $O/fuzz_real_$s.txt | cut -c1-600; done
