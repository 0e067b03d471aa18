#!/bin/bash
# round 5, call i: more output buffers per strand (what can be put aside while the allocator has the file), whole-buffer copies again
O=gpurun_out/r05i; mkdir -p $O
READS=64000000 REPS=3 MATRIX="FXH_ONE_FILE=0,,FXH_STRAND_OUT_SLOTS=2,FXH_STRAND_OUT_SLOTS=6,FXH_STRAND_OUT_SLOTS=8,FXH_STRAND_OUT_SLOTS=8:FXH_ONE_FILE_WINDOW_MB=256,FXH_STRAND_OUT_SLOTS=8:FXH_ONE_FILE_WINDOW_MB=512,FXH_STRAND_OUT_SLOTS=8:FXH_COPY_THREADS=32,FXH_STRAND_OUT_SLOTS=8:FXH_STRAND_MB=8,FXH_STRAND_OUT_SLOTS=6:FXH_STRANDS=6" timeout 900 python scripts/e2e_one_file.py > $O/e2e_one_file_64m.txt 2>&1
cat $O/e2e_one_file_64m.txt
