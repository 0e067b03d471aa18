#!/bin/bash
# round 5, call y: a tool that keeps little through the one-file run -- the head start's surplus pages given back during the run against kept until the end;
# then the headline tool, unchanged
O=gpurun_out/r05y; mkdir -p $O
READS=64000000 REPS=3 TOOL=fastq_quality_filter TOOL_ARGS="-q 36 -p 70" MATRIX="FXH_ONE_FILE=0,,FXH_ONE_FILE_KEEP_SURPLUS=1" timeout 600 python scripts/e2e_one_file.py > $O/e2e_keeps_little.txt 2>&1
grep -v "fxh timing strand" $O/e2e_keeps_little.txt | cut -c1-900
READS=64000000 REPS=3 MATRIX="FXH_ONE_FILE=0," timeout 600 python scripts/e2e_one_file.py > $O/e2e_headline_tool.txt 2>&1
grep -v "fxh timing strand" $O/e2e_headline_tool.txt | cut -c1-900
