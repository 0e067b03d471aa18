#!/bin/bash
# round-2 GPU call H: full GPU test tier, bench line, per-config rocprofv3 kernel statistics, PMC traffic of the cfg2 kernel
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
O=$R/gpurun_out/r02h; mkdir -p $O
export TMPDIR=/tmp
make -s -C fastx_toolkit_amd/host 2>/dev/null
if [ "${SKIP_TESTS:-0}" != "1" ]; then
echo "== gpu tests"
timeout 1500 python -m pytest tests -m gpu -q --durations=5 2>&1 | tail -12 | tee $O/pytest.txt
fi
echo "== bench (default)"
timeout 900 python bench.py --steps 20 --warmup 3 2>&1 | grep -v amdgpu.ids | tee $O/bench_cfg2.json | cut -c1-3000
echo "== per-config bench lines + rocprofv3 kernel statistics"
cd /tmp
for c in cfg2 cfg3 cfg4 cfg5shard stats; do
  rm -rf $R/gpurun_out/prof_$c
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$c -o bench -- python $R/bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline --no-e2e > $O/bench_${c}_under_rocprof.json 2> $O/prof_$c.err
  echo "rocprof $c rc=$?"
  db=$(find $R/gpurun_out/prof_$c -name "*.db" | head -1)
  if [ -n "$db" ]; then python $R/scripts/rocpd_stats.py $db $O/${c}_kernel_stats.md | head -6; fi
  for f in $(find $R/gpurun_out/prof_$c -name "*kernel_stats.csv"); do cp $f $O/${c}_kernel_stats.csv; done
  grep -v amdgpu.ids $O/bench_${c}_under_rocprof.json | cut -c1-400
  if [ "$c" != "cfg2" ]; then timeout 400 python $R/bench.py --config $c --steps 10 --warmup 2 2>&1 | grep -v amdgpu.ids > $O/bench_$c.json; cut -c1-600 $O/bench_$c.json; fi
  rm -rf $R/gpurun_out/prof_$c
done
echo "== PMC traffic (cfg2 kernel)"
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc/$ctr
  timeout 400 rocprofv3 --pmc $ctr -d $R/gpurun_out/pmc/$ctr -o pmc --output-format csv -- python $R/scripts/pmc_run.py > $O/pmc_$ctr.log 2>&1
  echo "pmc $ctr rc=$?"; tail -2 $O/pmc_$ctr.log
done
python $R/scripts/pmc_parse.py $R/gpurun_out/pmc/FETCH_SIZE $R/gpurun_out/pmc/WRITE_SIZE 2>&1 | grep -i "tiles\|elementwise\|copy" | head -12
