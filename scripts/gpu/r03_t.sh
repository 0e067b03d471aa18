#!/bin/bash
# round-3 GPU call T: clip kernel per adapter length; the bench lines of cfg3 / cfg4 / stats with the committed PMC traffic attached
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r03t; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python scripts/clip_by_adapter_len.py > $O/clip_by_adapter_len.txt 2> $O/clip_by_adapter_len.err; echo "adapter rc=$?"; cat $O/clip_by_adapter_len.txt; tail -3 $O/clip_by_adapter_len.err
for c in cfg3 cfg4 stats; do
  timeout 600 python bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err; echo "bench $c rc=$?"
  python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r03t/bench_$c.json") if l.startswith("{")][-1])
r = d["roofline"]
print("$c", d["value"], d["ms_per_step"], r["frac"], r.get("traffic"), r.get("traffic_over_algorithmic"), r.get("traffic_source"), d.get("self_check"))
PY
done
