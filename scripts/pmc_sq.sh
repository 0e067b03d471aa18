#!/bin/bash
# SQ counters of one workload script (one counter group per rocprofv3 pass, counters only -- no trace domains).
# usage: pmc_sq.sh <outdir-under-gpurun_out> <python script> ; env passes through
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; SCRIPT=$2
cd /tmp && export TMPDIR=/tmp
mkdir -p $OUT
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"; do
  d=$OUT/$(echo $grp | tr ' ' '_' | cut -c1-50)
  rm -rf $d
  timeout 300 rocprofv3 --pmc $grp -d $d -o pmc --output-format csv -- python $R/$SCRIPT > $d.log 2>&1
  echo "pmc rc=$? $(tail -1 $d.log | cut -c1-100)"
done
python $R/scripts/pmc_parse.py $OUT/*/ 2>/dev/null | grep -v "memset\|elementwise\|synth\|finish_counters" > $OUT/summary.txt
cat $OUT/summary.txt | cut -c1-160
