#!/bin/bash
# round-4 closing evidence call at HEAD: (1) FETCH_SIZE / WRITE_SIZE passes of every config's dominant kernel and the traffic files made
# from them ON the box, so that (2) the bench lines of the same call carry roofline.traffic; per config the rocprofv3 kernel statistics
# of the same command; (3) the default bench line (cpu_baseline, e2e legs) and the cfg5 shard line with the rank-parallel tool chain.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
O=$R/gpurun_out/r04final; mkdir -p $O/profiles
export TMPDIR=/tmp
CFGS="cfg2 cfg3 cfg4 cfg5shard stats"
bash scripts/gpu/r03_pmc.sh r04final/pmc "$CFGS"
for c in $CFGS; do
  python scripts/pmc_traffic.py gpurun_out/r04final/pmc/$c r04 $c > $O/pmc_traffic_$c.log 2>&1; echo "traffic $c rc=$? $(grep traffic_over_algorithmic $O/pmc_traffic_$c.log)"
  cp profiles/pmc_traffic_$c.json $O/profiles/; rm -rf $O/profiles/r04_pmc_$c; cp -r profiles/r04_pmc_$c $O/profiles/
done
cd /tmp
for c in $CFGS; do
  rm -rf $R/gpurun_out/prof_$c
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$c -o bench -- python $R/bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline --no-e2e > $O/${c}_bench_under_rocprof.json 2> $O/prof_$c.err
  echo "rocprof $c rc=$?"
  db=$(find $R/gpurun_out/prof_$c -name "*.db" | head -1)
  if [ -n "$db" ]; then python $R/scripts/rocpd_stats.py $db $O/${c}_kernel_stats.md | head -3; fi
  rm -rf $R/gpurun_out/prof_$c
  if [ "$c" != "cfg2" ]; then
    timeout 500 python $R/bench.py --config $c --steps 10 --warmup 2 --no-e2e 2>&1 | grep "^{" > $O/${c}_bench.json; cut -c1-140 $O/${c}_bench.json
  fi
done
cd $R
timeout 900 python bench.py 2> $O/bench_default.err | grep "^{" > $O/cfg2_bench.json; echo "default bench rc=$?"; cut -c1-300 $O/cfg2_bench.json
timeout 600 python bench.py --config cfg5shard --e2e --no-cpu-baseline 2> $O/bench_cfg5_e2e.err | grep "^{" > $O/cfg5shard_e2e_bench.json; echo "cfg5 e2e rc=$?"
python - <<'PY'
import json
for f in ("cfg2_bench.json", "cfg5shard_e2e_bench.json"):
    try:
        d = json.loads(open("gpurun_out/r04final/" + f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f, d["value"], d["ms_per_step"], r["frac"], r.get("traffic_over_algorithmic"), d.get("self_check", {}).get("matches_pinned"), json.dumps(d.get("e2e_ranks"))[:400], json.dumps((d.get("e2e") or {}).get("sharded_big"))[:300])
    except Exception as e:
        print(f, "unreadable", e)
PY
