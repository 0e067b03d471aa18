#!/bin/bash
# round 5, call h: eager allocator again, copies a megabyte at a time under the gate, NUMA node from sysfs, 16 MB chunks
O=gpurun_out/r05h; mkdir -p $O
READS=64000000 REPS=3 MATRIX="FXH_ONE_FILE=0,,FXH_STRANDS=6,FXH_STRANDS=10,FXH_ONE_FILE_WINDOW_MB=64,FXH_ONE_FILE_WINDOW_MB=256,FXH_STRAND_MB=8,FXH_STRAND_MB=24,FXH_STRAND_READERS=3" timeout 900 python scripts/e2e_one_file.py > $O/e2e_one_file_64m.txt 2>&1
cat $O/e2e_one_file_64m.txt
