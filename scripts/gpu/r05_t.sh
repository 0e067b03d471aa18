#!/bin/bash
# round 5, call t: the CLI campaign over the REAL engine (one-file strands, rank jobs of 2-4 processes on the one GPU through the test transport, lanes, parts) and the CLI tier
O=gpurun_out/r05t; mkdir -p $O
FXG_CAMPAIGN_REAL=1 timeout 400 python scripts/fuzz_campaign_cli.py 601 150 > $O/fuzz_real_601.txt 2>&1 &
FXG_CAMPAIGN_REAL=1 timeout 400 python scripts/fuzz_campaign_cli.py 602 150 > $O/fuzz_real_602.txt 2>&1 &
wait
timeout 900 python -m pytest tests/test_gpu_cli.py -q -m gpu > $O/pytest_cli.txt 2>&1
tail -3 $O/fuzz_real_601.txt $O/fuzz_real_602.txt; tail -5 $O/pytest_cli.txt
