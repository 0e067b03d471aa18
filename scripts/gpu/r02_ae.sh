#!/bin/bash
# round-2 GPU call AE: the per-config bench lines and rocprofv3 kernel statistics again at the end of the round (cfg3, cfg4, cfg5shard, stats)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
O=$R/gpurun_out/r02ae; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for c in cfg3 cfg4 cfg5shard stats; do
  rm -rf $R/gpurun_out/prof_$c
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$c -o bench -- python $R/bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline --no-e2e > $O/bench_${c}_under_rocprof.json 2> $O/prof_$c.err
  echo "rocprof $c rc=$?"
  db=$(find $R/gpurun_out/prof_$c -name "*.db" | head -1)
  if [ -n "$db" ]; then python $R/scripts/rocpd_stats.py $db $O/${c}_kernel_stats.md | head -4; fi
  timeout 400 python $R/bench.py --config $c --steps 10 --warmup 2 --no-e2e 2>&1 | grep -v amdgpu.ids > $O/bench_$c.json; cut -c1-200 $O/bench_$c.json
  rm -rf $R/gpurun_out/prof_$c
done
