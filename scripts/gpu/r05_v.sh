#!/bin/bash
# round 5, call v: the cfg2 kernel statistics once more with the headline config alone (the closing call's cfg2 pass had the other configs behind it)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; O=$R/gpurun_out/r05v; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
rm -rf $R/gpurun_out/prof_cfg2
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_cfg2 -o bench -- python $R/bench.py --config cfg2 --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --headline-only > $O/cfg2_bench_under_rocprof.json 2> $O/prof_cfg2.err
db=$(find $R/gpurun_out/prof_cfg2 -name "*.db" | head -1); python $R/scripts/rocpd_stats.py $db $O/cfg2_kernel_stats.md | head -4
rm -rf $R/gpurun_out/prof_cfg2
