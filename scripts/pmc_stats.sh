#!/bin/bash
# SQ counters of fxg_kernel_quality_stats (one counter group per pass, counters only). Output: gpurun_out/pmc_stats/<group>/
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU" "SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ATOMIC_RETURN SQ_ACTIVE_INST_ANY"; do
  d=$R/gpurun_out/pmc_stats/$(echo $grp | tr ' ' '_' | cut -c1-60)
  rm -rf $d; mkdir -p $R/gpurun_out/pmc_stats
  timeout 200 rocprofv3 --pmc $grp -d $d -o pmc --output-format csv -- python $R/scripts/bench_stats.py > $d.log 2>&1
  tail -1 $d.log | cut -c1-100
done
