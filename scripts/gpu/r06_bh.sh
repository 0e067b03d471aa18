#!/bin/bash
# round 6, call bh: the command-line campaign over the REAL engine at the final HEAD (random tools, flags and inputs against the real libfastx driver), seeds $2.. for $1 s each
O=gpurun_out/r06bh; mkdir -p $O
for seed in $(( ${2:-901} )) $(( ${2:-901} + 1 )); do FXG_CAMPAIGN_REAL=1 timeout $(( $1 + 300 )) python scripts/fuzz_campaign_cli.py $seed $1 2>&1 | tail -n 3 | cut -c1-400; done | tee $O/fuzz_campaign_cli_real.txt
CASES=3 timeout 600 python scripts/gather_alignment.py 2>&1 | grep "^{" | tee $O/gather_alignment_rev_trim.txt | cut -c1-200
