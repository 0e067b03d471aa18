#!/bin/bash
# round 5, call j: where the 50 ms of placement and the 130 ms after the last chunk go
O=gpurun_out/r05j; mkdir -p $O
READS=64000000 REPS=2 MATRIX="FXH_ONE_FILE=0,,FXH_NO_NUMA=1,FXH_TEARDOWN=1,FXH_STRANDS=4" timeout 900 python scripts/e2e_one_file.py > $O/e2e_one_file_64m.txt 2>&1
cat $O/e2e_one_file_64m.txt
