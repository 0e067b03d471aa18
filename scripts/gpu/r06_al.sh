#!/bin/bash
# round 6, call al: random campaign of the kernels through the C-ABI against the oracle (scripts/fuzz_campaign_gpu.py), seeds $2, $2 + 1, $2 + 2 (default 601..) for $1 seconds each
O=gpurun_out/r06al; mkdir -p $O
for seed in $(( ${2:-601} )) $(( ${2:-601} + 1 )) $(( ${2:-601} + 2 )); do timeout $(( $1 + 120 )) python scripts/fuzz_campaign_gpu.py $seed $1 2>&1 | tail -n 4 | cut -c1-600; done | tee $O/fuzz_campaign_gpu.txt
