#!/bin/bash
# round 6, call p: the whole GPU tier (clip matrix included) at HEAD, smoke, and the default bench line with the traffic files of this round's kernels in place
O=gpurun_out/r06p; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 1 $O/smoke.log
timeout 3000 python -m pytest tests -q -m gpu --durations=6 > $O/pytest_gpu.txt 2>&1; tail -n 12 $O/pytest_gpu.txt
timeout 1500 python bench.py 2> $O/bench_default.err | grep "^{" > $O/bench_default.json; echo "default bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06p/bench_default.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("cfg2", d["value"], d["ms_per_step"], r["frac"], r.get("traffic_over_algorithmic"), d.get("self_check", {}).get("matches_pinned"))
for k, v in (r.get("other_configs") or {}).items():
    print("   ", k, {x: v.get(x) for x in ("mreads_s", "kernel_ms_avg", "frac", "issued_frac", "hbm_frac", "traffic_over_algorithmic", "self_check_matches_pinned", "cpu_baseline_mreads_s")})
PY
