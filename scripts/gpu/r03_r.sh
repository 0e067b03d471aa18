#!/bin/bash
# round-3 GPU call R: whole GPU tier, then the bench lines (default = cfg2 with cpu_baseline and the e2e legs; cfg5shard with --e2e)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
O=$R/gpurun_out/r03r; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu --durations=6 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest_gpu.log
timeout 600 python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err; echo "bench rc=$?"; cut -c1-1500 $O/bench_cfg2.json; tail -3 $O/bench_cfg2.err
timeout 600 python bench.py --config cfg5shard --e2e --no-cpu-baseline > $O/bench_cfg5shard_e2e.json 2> $O/bench_cfg5.err; echo "bench cfg5 rc=$?"; python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r03r/bench_cfg5shard_e2e.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["roofline"]["valu"], d.get("self_check"), d.get("e2e_ranks"))
PY
