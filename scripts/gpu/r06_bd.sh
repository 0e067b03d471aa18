#!/bin/bash
# round 6, call bd: the kernel campaign on the GPU at the final HEAD (the statistics kernel's piece form and the bitmap census among what it draws), seeds $2.. for $1 s each
O=gpurun_out/r06bd; mkdir -p $O
for seed in $(( ${2:-801} )) $(( ${2:-801} + 1 )) $(( ${2:-801} + 2 )); do timeout $(( $1 + 120 )) python scripts/fuzz_campaign_gpu.py $seed $1 2>&1 | tail -n 2 | cut -c1-600; done | tee $O/fuzz_campaign_gpu.txt
