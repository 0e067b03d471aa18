#!/bin/bash
# round 5, call x: rank jobs into ONE tmpfs file -- positional writes by every rank against copies into pages rank 0 made while the ranks computed;
# the sink alone first (scripts/ubench/rank_drain.c), then the tool: 2 and 4 ranks sharing the box's one GPU
O=gpurun_out/r05x; mkdir -p $O
gcc -O2 -pthread scripts/ubench/rank_drain.c -o /tmp/rank_drain
{ for m in w x a; do for pt in "8 2" "8 4" "4 4"; do /tmp/rank_drain $m /dev/shm 20 $pt; done; done; } > $O/rank_drain.txt 2>&1
cat $O/rank_drain.txt
READS=64000000 REPS=2 MATRIX="FXH_ONE_FILE=0,,RANKS=2:FXH_ONE_FILE_SINK=pwrite,RANKS=2,RANKS=4:FXH_ONE_FILE_SINK=pwrite,RANKS=4,FXH_RANK_MODE=1:FXH_ONE_FILE_SINK=pwrite,FXH_RANK_MODE=1" timeout 900 python scripts/e2e_one_file.py > $O/e2e_rank_jobs.txt 2>&1
grep -v "fxh timing strand\|^    fxh timing one file" $O/e2e_rank_jobs.txt | cut -c1-330
