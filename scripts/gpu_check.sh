#!/bin/bash
# Runs on the GPU box (via gpurun): smoke -> pytest -m gpu -> bench -> rocprofv3 kernel stats.
# Everything is wrapped in timeout; outputs land in gpurun_out/ (merged back by gpurun).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/gpu.txt
nproc >> gpurun_out/gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"
tail -3 gpurun_out/smoke.log
if [ "${SKIP_TESTS:-0}" != "1" ]; then
timeout 900 python -m pytest tests -x -q -m gpu --durations=12 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -25 gpurun_out/pytest_gpu.log
fi
timeout 600 python bench.py --steps ${STEPS:-20} --warmup 3 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"
cat gpurun_out/bench.json; tail -5 gpurun_out/bench.err
if [ "${SKIP_PROF:-0}" != "1" ]; then
cd /tmp
rm -rf $R/gpurun_out/prof
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o bench -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e > $R/gpurun_out/bench_prof.json 2> $R/gpurun_out/prof.err; echo "rocprof rc=$?"
cd $R
find gpurun_out/prof -name "*stats*" | head; 
for f in $(find gpurun_out/prof -name "*kernel_stats.csv"); do head -12 $f; done
# keep the merged-back payload small: drop the raw per-dispatch trace, keep the stats
find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
fi
