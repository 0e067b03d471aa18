#!/bin/bash
# round 6, call bp: the command-line campaign over the real engine at the round's last kernel sources, seeds $2.. for $1 s each
O=gpurun_out/r06bp; mkdir -p $O
for seed in $(( ${2:-961} )) $(( ${2:-961} + 1 )); do FXG_CAMPAIGN_REAL=1 timeout $(( $1 + 300 )) python scripts/fuzz_campaign_cli.py $seed $1 2>&1 | tail -n 3 | cut -c1-400; done | tee $O/fuzz_campaign_cli_real.txt
