#!/bin/bash
# round 5, call g: copies-first gate; what finding the GPU's NUMA node costs without the runtime
O=gpurun_out/r05g; mkdir -p $O
./scripts/ubench/sysfs_timing > $O/sysfs_timing.txt 2>&1
READS=64000000 REPS=3 MATRIX="FXH_ONE_FILE=0,,FXH_STRAND_MB=16,FXH_STRAND_MB=16:FXH_STRANDS=12,FXH_STRAND_MB=16:FXH_ONE_FILE_IDLE_WINDOW_MB=8,FXH_STRAND_MB=16:FXH_ONE_FILE_IDLE_WINDOW_MB=128,FXH_STRAND_MB=16:FXH_ONE_FILE_WINDOW_MB=256,FXH_STRAND_MB=16:FXH_STRAND_READERS=4,FXH_STRAND_MB=16:FXH_COPY_THREADS=32:FXH_STRANDS=12" timeout 900 python scripts/e2e_one_file.py > $O/e2e_one_file_64m.txt 2>&1
cat $O/sysfs_timing.txt; cat $O/e2e_one_file_64m.txt
