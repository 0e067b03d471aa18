#!/bin/bash
# round 5, call s: four strands x four readers as the default; a tool that keeps little of its input (input-bound: do four strands still fill the link?)
O=gpurun_out/r05s; mkdir -p $O
READS=64000000 REPS=3 MATRIX="FXH_ONE_FILE=0,,FXH_STRANDS=8:FXH_STRAND_READERS=2,FXH_STRAND_READERS=2" timeout 900 python scripts/e2e_one_file.py > $O/e2e_one_file_64m.txt 2>&1
READS=64000000 REPS=3 TOOL=fastq_quality_filter TOOL_ARGS="-q 30 -p 90" MATRIX="FXH_ONE_FILE=0,,FXH_STRANDS=8:FXH_STRAND_READERS=2,FXH_STRANDS=6" timeout 900 python scripts/e2e_one_file.py > $O/e2e_one_file_64m_filter_keeps_little.txt 2>&1
grep -v "timing lane\|timing part\|timing exit\|placement:" $O/e2e_one_file_64m.txt $O/e2e_one_file_64m_filter_keeps_little.txt | cut -c1-330
