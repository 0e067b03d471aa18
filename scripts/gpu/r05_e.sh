#!/bin/bash
# round 5, call e: the default command line into ONE file, first runs of fxh_strands.c on the real engine
O=gpurun_out/r05e; mkdir -p $O
READS=64000000 REPS=2 MATRIX="FXH_ONE_FILE=0,,FXH_STRANDS=4,FXH_STRANDS=6,FXH_STRANDS=12,FXH_STRAND_MB=16,FXH_STRAND_MB=4:FXH_STRANDS=12,FXH_STRAND_READERS=4,FXH_ONE_FILE_SINK=pwrite,FXH_ONE_FILE_WINDOW_MB=256,FXH_COPY_THREADS=32" timeout 900 python scripts/e2e_one_file.py > $O/e2e_one_file_64m.txt 2>&1
READS=16000000 REPS=2 MATRIX="FXH_ONE_FILE=0,," timeout 300 python scripts/e2e_one_file.py > $O/e2e_one_file_16m.txt 2>&1
cat $O/e2e_one_file_64m.txt $O/e2e_one_file_16m.txt
