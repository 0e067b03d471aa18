#!/bin/bash
# round-2 GPU call AA: final state with fxg_kernel_rows as the cfg2 kernel -- full GPU test tier, smoke, bench lines of both cfg2
# kernels, rocprofv3 kernel statistics, PMC traffic, the mixed read/write streaming microbenchmark
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
O=$R/gpurun_out/r02aa; mkdir -p $O
export TMPDIR=/tmp
make -s -C fastx_toolkit_amd/host 2>/dev/null
echo "== gpu tests"
timeout 1500 python -m pytest tests -m gpu -q --durations=5 2>&1 | tail -12 | tee $O/pytest.txt
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -3 | tee $O/smoke.txt
echo "== bench (default = rows), then the tile kernel, twice each"
for i in 1 2; do
timeout 900 python bench.py --steps 20 --warmup 3 2>&1 | grep -v amdgpu.ids | tee $O/bench_cfg2_$i.json | cut -c1-200
FXG_ROWS=0 timeout 900 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-e2e 2>&1 | grep -v amdgpu.ids | tee $O/bench_cfg2_tiles_$i.json | cut -c1-200
done
echo "== rocprofv3 kernel statistics (cfg2)"
cd /tmp
rm -rf $R/gpurun_out/prof_cfg2
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_cfg2 -o bench -- python $R/bench.py --config cfg2 --steps 10 --warmup 2 --no-cpu-baseline --no-e2e > $O/bench_cfg2_under_rocprof.json 2> $O/prof_cfg2.err
echo "rocprof rc=$?"
db=$(find $R/gpurun_out/prof_cfg2 -name "*.db" | head -1)
if [ -n "$db" ]; then python $R/scripts/rocpd_stats.py $db $O/cfg2_kernel_stats.md | head -6; fi
for f in $(find $R/gpurun_out/prof_cfg2 -name "*kernel_stats.csv"); do cp $f $O/cfg2_kernel_stats.csv; done
rm -rf $R/gpurun_out/prof_cfg2
echo "== PMC traffic (cfg2 kernel)"
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc/$ctr
  timeout 400 rocprofv3 --pmc $ctr -d $R/gpurun_out/pmc/$ctr -o pmc --output-format csv -- python $R/scripts/pmc_run.py > $O/pmc_$ctr.log 2>&1
  echo "pmc $ctr rc=$?"
done
python $R/scripts/pmc_parse.py $R/gpurun_out/pmc/FETCH_SIZE $R/gpurun_out/pmc/WRITE_SIZE 2>&1 | grep -i "rows\|tiles\|copy" | head -8
echo "== mixed read/write streaming microbenchmark"
cd $R; timeout 300 scripts/ubench/mix_rw | tee $O/mix_rw.txt
