#!/bin/bash
# round 5, call f: the one-file run's time line, chunk sizes and strand counts
O=gpurun_out/r05f; mkdir -p $O
READS=64000000 REPS=2 STRAND_LINES=1 MATRIX="FXH_ONE_FILE=0,,FXH_STRAND_MB=16,FXH_STRAND_MB=32,FXH_STRAND_MB=16:FXH_STRANDS=12,FXH_STRAND_MB=16:FXH_STRANDS=6,FXH_STRAND_MB=16:FXH_STRAND_READERS=4,FXH_STRAND_MB=16:FXH_ONE_FILE_WINDOW_MB=64,FXH_STRAND_MB=16:FXH_ONE_FILE_WINDOW_MB=512" timeout 900 python scripts/e2e_one_file.py > $O/e2e_one_file_64m.txt 2>&1
grep -v "timing strand [1-9]" $O/e2e_one_file_64m.txt
