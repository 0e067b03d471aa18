#!/bin/bash
# round-6 closing evidence call at HEAD: (1) FETCH_SIZE / WRITE_SIZE passes of every config's dominant kernel and SQ_INSTS_VALU of the clip configs, and the
# files made from them ON the box, so that (2) the bench lines of the same call carry roofline.traffic / issued_*; per config the rocprofv3 kernel
# statistics of the same command; (3) the default bench line (all configs inside roofline.other_configs, cpu baselines, e2e legs).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
O=$R/gpurun_out/r06final; mkdir -p $O/profiles
export TMPDIR=/tmp
CFGS="cfg2 cfg3 cfg4 cfg5shard stats"
bash scripts/gpu/r03_pmc.sh r06final/pmc "$CFGS"
for c in $CFGS; do
  python scripts/pmc_traffic.py gpurun_out/r06final/pmc/$c r06 $c > $O/pmc_traffic_$c.log 2>&1; echo "traffic $c rc=$? $(grep traffic_over_algorithmic $O/pmc_traffic_$c.log)"
  cp profiles/pmc_traffic_$c.json $O/profiles/; rm -rf $O/profiles/r06_pmc_$c; cp -r profiles/r06_pmc_$c $O/profiles/
done
cd /tmp
for c in cfg3 cfg5shard; do
  d=$R/gpurun_out/r06final/pmc/$c/SQ_INSTS_VALU; rm -rf $d
  CFG=$c timeout 300 rocprofv3 --pmc SQ_INSTS_VALU -d $d -o pmc --output-format csv -- python $R/scripts/pmc_run.py > $d.log 2>&1
  f=$(find $d -name "*counter_collection.csv" | head -1); [ -n "$f" ] && [ "$f" != "$d/pmc_counter_collection.csv" ] && mv $f $d/pmc_counter_collection.csv
  (cd $R && python scripts/pmc_sq_json.py gpurun_out/r06final/pmc/$c/SQ_INSTS_VALU r06 $c | cut -c1-300; cp profiles/pmc_sq_$c.json $O/profiles/; cp -r profiles/r06_pmc_sq_$c $O/profiles/)
done
for c in $CFGS; do
  rm -rf $R/gpurun_out/prof_$c
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$c -o bench -- python $R/bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --headline-only > $O/${c}_bench_under_rocprof.json 2> $O/prof_$c.err
  echo "rocprof $c rc=$?"
  db=$(find $R/gpurun_out/prof_$c -name "*.db" | head -1)
  if [ -n "$db" ]; then python $R/scripts/rocpd_stats.py $db $O/${c}_kernel_stats.md | head -3; fi
  rm -rf $R/gpurun_out/prof_$c
done
cd $R
timeout 1500 python bench.py 2> $O/bench_default.err | grep "^{" > $O/cfg2_bench.json; echo "default bench rc=$?"; cut -c1-300 $O/cfg2_bench.json
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06final/cfg2_bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("cfg2", d["value"], d["ms_per_step"], r["frac"], r.get("traffic_over_algorithmic"), d.get("self_check", {}).get("matches_pinned"))
for k, v in (r.get("other_configs") or {}).items():
    print("   ", k, {x: v.get(x) for x in ("mreads_s", "ms_per_step", "kernel_ms_avg", "frac", "issued_frac", "gcups", "hbm_frac", "traffic_over_algorithmic", "self_check_matches_pinned", "cpu_baseline_mreads_s", "cpu_baseline_cores")})
e = d.get("e2e") or {}
print("    e2e", {k: (v.get("mreads_s") if isinstance(v, dict) else v) for k, v in e.items() if k not in ("sharded_big", "default_invocation")}, {k: (v.get("mreads_s") if isinstance(v, dict) else v) for k, v in e.get("sharded_big", {}).items()})
PY
