#!/bin/bash
# round 5, call l: from which input size the many-strand run pays; NUMA node from the topology files
O=gpurun_out/r05l; mkdir -p $O
for r in 1000000 2000000 4000000 8000000 16000000; do
  READS=$r REPS=3 MATRIX="FXH_ONE_FILE=0,FXH_ONE_FILE_MIN_MB=0,FXH_ONE_FILE_MIN_MB=0:FXH_STRANDS=4,FXH_ONE_FILE_MIN_MB=0:FXH_STRAND_MB=8" timeout 300 python scripts/e2e_one_file.py
done > $O/e2e_one_file_by_size.txt 2>&1
READS=64000000 REPS=3 MATRIX="FXH_ONE_FILE=0," timeout 600 python scripts/e2e_one_file.py > $O/e2e_one_file_64m.txt 2>&1
grep -v "timing lane\|timing part\|timing exit\|child was gone" $O/e2e_one_file_by_size.txt $O/e2e_one_file_64m.txt
