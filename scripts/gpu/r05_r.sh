#!/bin/bash
# round 5, call r: the one-file run with the allocator started ahead of the placement query; strands 4 / 8; 3 repeats each
O=gpurun_out/r05r; mkdir -p $O
READS=64000000 REPS=4 MATRIX="FXH_ONE_FILE=0,,FXH_STRANDS=4,FXH_STRANDS=6" timeout 900 python scripts/e2e_one_file.py > $O/e2e_one_file_64m.txt 2>&1
grep -v "timing lane\|timing part\|timing exit" $O/e2e_one_file_64m.txt
