#!/bin/bash
# round-2 GPU call X: HBM traffic of fxg_kernel_rows (PMC passes, counters only)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/r02x; mkdir -p $O
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc/$ctr
  timeout 400 rocprofv3 --pmc $ctr -d $R/gpurun_out/pmc/$ctr -o pmc --output-format csv -- python $R/scripts/pmc_run.py > $O/pmc_$ctr.log 2>&1
  echo "pmc $ctr rc=$?"; tail -2 $O/pmc_$ctr.log
done
python $R/scripts/pmc_parse.py $R/gpurun_out/pmc/FETCH_SIZE $R/gpurun_out/pmc/WRITE_SIZE 2>&1 | grep -i "rows\|tiles\|elementwise\|copy" | head -12 | tee $O/pmc.txt
