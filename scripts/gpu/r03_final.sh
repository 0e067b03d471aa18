#!/bin/bash
# round-3 measurement call: per config the bench line (HIP events) and the rocprofv3 kernel statistics of the same command; the PMC
# traffic passes; the SQ counters of the clip kernel.  Everything lands under gpurun_out/r03final/ and is copied to profiles/ by hand.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
O=$R/gpurun_out/r03final; mkdir -p $O
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -6 > $O/gpu.txt; nproc >> $O/gpu.txt
cd /tmp
for c in ${CFGS:-cfg2 cfg3 cfg4 cfg5shard stats}; do
  rm -rf $R/gpurun_out/prof_$c
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$c -o bench -- python $R/bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline --no-e2e > $O/${c}_bench_under_rocprof.json 2> $O/prof_$c.err
  echo "rocprof $c rc=$?"
  db=$(find $R/gpurun_out/prof_$c -name "*.db" | head -1)
  if [ -n "$db" ]; then python $R/scripts/rocpd_stats.py $db $O/${c}_kernel_stats.md | head -4; fi
  extra=""; [ "$c" != "cfg2" ] && extra="--no-e2e"
  timeout 500 python $R/bench.py --config $c --steps 10 --warmup 2 $extra 2>&1 | grep -v amdgpu.ids | grep "^{" > $O/${c}_bench.json; cut -c1-160 $O/${c}_bench.json
  rm -rf $R/gpurun_out/prof_$c
done
cd $R
bash scripts/gpu/r03_pmc.sh r03final/pmc "${CFGS:-cfg2 cfg3 cfg4 cfg5shard stats}"
CFG=cfg3 bash scripts/pmc_sq.sh r03final/pmc_sq_cfg3 scripts/pmc_clip.py
CFG=cfg5 bash scripts/pmc_sq.sh r03final/pmc_sq_cfg5 scripts/pmc_clip.py
