#!/bin/bash
# round 4, call k: new CLI tests on the real runtime (abandon path last), bench lines with the new roofline shape, un-tuned command lines
mkdir -p gpurun_out/r04k
timeout 900 python -m pytest tests/test_gpu_cli.py -q -m gpu -k "abandoned or default_command_line or sharded_run_and" > gpurun_out/r04k/pytest_cli.txt 2>&1; tail -5 gpurun_out/r04k/pytest_cli.txt
timeout 600 python bench.py --config cfg3 --no-e2e > gpurun_out/r04k/bench_cfg3.json 2> gpurun_out/r04k/bench_cfg3.err; tail -c 600 gpurun_out/r04k/bench_cfg3.err
timeout 600 python bench.py --config stats > gpurun_out/r04k/bench_stats.json 2> gpurun_out/r04k/bench_stats.err; tail -c 300 gpurun_out/r04k/bench_stats.err
timeout 900 python scripts/e2e_defaults.py > gpurun_out/r04k/e2e_defaults.txt 2> gpurun_out/r04k/e2e_defaults.err; cat gpurun_out/r04k/e2e_defaults.txt; tail -c 300 gpurun_out/r04k/e2e_defaults.err
timeout 900 python bench.py > gpurun_out/r04k/bench_cfg2.json 2> gpurun_out/r04k/bench_cfg2.err; tail -c 300 gpurun_out/r04k/bench_cfg2.err
python - <<'PY'
import json
for c in ("cfg3","stats","cfg2"):
    try:
        d=json.loads(open("gpurun_out/r04k/bench_%s.json"%c).read().strip().splitlines()[-1])
        r=d["roofline"]; print(c, d["value"], d["ms_per_step"], r["bound"], r["frac"], r["kernel_ms_avg"], r.get("hbm",{}).get("frac"), (d.get("cpu_baseline") or {}).get("value"))
        if "e2e" in d: print(json.dumps({k:(v if not isinstance(v,dict) else {kk:vv for kk,vv in v.items() if kk in("mreads_s","wall_s","sharded","one_stream","sharded_host_share_16c")}) for k,v in d["e2e"].items()})[:1500])
    except Exception as e: print(c,"ERR",e)
PY
