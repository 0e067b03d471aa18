#!/bin/bash
# round 6, call ap: the command-line campaign over the REAL engine (FXG_CAMPAIGN_REAL=1): random tools, flags and inputs against the real libfastx driver, seeds $2.. for $1 s each;
# then two more seeds of the kernel campaign
O=gpurun_out/r06ap; mkdir -p $O
for seed in $(( ${2:-701} )) $(( ${2:-701} + 1 )); do FXG_CAMPAIGN_REAL=1 timeout $(( $1 + 300 )) python scripts/fuzz_campaign_cli.py $seed $1 2>&1 | tail -n 3 | cut -c1-400; done | tee $O/fuzz_campaign_cli_real.txt
for seed in $(( ${2:-701} + 10 )) $(( ${2:-701} + 11 )); do timeout $(( $1 + 120 )) python scripts/fuzz_campaign_gpu.py $seed $1 2>&1 | tail -n 2 | cut -c1-400; done | tee $O/fuzz_campaign_gpu.txt
