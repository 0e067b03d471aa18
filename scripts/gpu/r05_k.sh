#!/bin/bash
# round 5, call k: strands releasing their own device state side by side; what the topology files say about the GPU
O=gpurun_out/r05k; mkdir -p $O
./scripts/ubench/sysfs_timing > $O/sysfs_timing.txt 2>&1
READS=64000000 REPS=3 MATRIX="FXH_ONE_FILE=0,,FXH_STRAND_RELEASE=1,FXH_STRAND_RELEASE=1:FXH_STRANDS=6" timeout 900 python scripts/e2e_one_file.py > $O/e2e_one_file_64m.txt 2>&1
grep -v "0 bytes" $O/sysfs_timing.txt | head -60; cat $O/e2e_one_file_64m.txt
