#!/bin/bash
# round-5 closing evidence call at HEAD: (1) FETCH_SIZE / WRITE_SIZE passes of every config's dominant kernel and the traffic files made from them ON the
# box, so that (2) the bench lines of the same call carry roofline.traffic; per config the rocprofv3 kernel statistics of the same command; (3) the tool's
# device text path under rocprofv3 on 64 M reads (the default command line, one output file); (4) the default bench line (all configs, cpu baselines, e2e
# legs) and the cfg5 line with the tools' rank mode.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
O=$R/gpurun_out/r05final; mkdir -p $O/profiles
export TMPDIR=/tmp
CFGS="cfg2 cfg3 cfg4 cfg5shard stats"
bash scripts/gpu/r03_pmc.sh r05final/pmc "$CFGS"
for c in $CFGS; do
  python scripts/pmc_traffic.py gpurun_out/r05final/pmc/$c r05 $c > $O/pmc_traffic_$c.log 2>&1; echo "traffic $c rc=$? $(grep traffic_over_algorithmic $O/pmc_traffic_$c.log)"
  cp profiles/pmc_traffic_$c.json $O/profiles/; rm -rf $O/profiles/r05_pmc_$c; cp -r profiles/r05_pmc_$c $O/profiles/
done
cd /tmp
for c in $CFGS; do
  rm -rf $R/gpurun_out/prof_$c
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$c -o bench -- python $R/bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --headline-only > $O/${c}_bench_under_rocprof.json 2> $O/prof_$c.err
  echo "rocprof $c rc=$?"
  db=$(find $R/gpurun_out/prof_$c -name "*.db" | head -1)
  if [ -n "$db" ]; then python $R/scripts/rocpd_stats.py $db $O/${c}_kernel_stats.md | head -3; fi
  rm -rf $R/gpurun_out/prof_$c
done
# (3) the text path: generate 64 M reads once, run the default command line under the kernel trace
python - <<'PY' > $O/textpath_gen.log 2>&1
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
bench._gen_fastq("/dev/shm/r05_in.fq", 64_000_000)
print(os.path.getsize("/dev/shm/r05_in.fq"))
PY
rm -rf $R/gpurun_out/prof_text
FXH_SLOW_EXIT=1 timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_text -o tool -- $R/fastx_toolkit_amd/host/bin/fastq_quality_trim_filter -t 20 -l 30 -q 20 -p 80 -i /dev/shm/r05_in.fq -o /dev/shm/r05_out.fq > $O/textpath_tool.log 2>&1
echo "rocprof text path rc=$?"
for db in $(find $R/gpurun_out/prof_text -name "*.db"); do python $R/scripts/rocpd_stats.py $db $O/textpath_kernel_stats_$(basename $db .db).md | head -3; done
ls -la /dev/shm/r05_out.fq | awk '{print "output bytes", $5}'
rm -rf $R/gpurun_out/prof_text /dev/shm/r05_in.fq /dev/shm/r05_out.fq
cd $R
timeout 1500 python bench.py 2> $O/bench_default.err | grep "^{" > $O/cfg2_bench.json; echo "default bench rc=$?"; cut -c1-300 $O/cfg2_bench.json
timeout 600 python bench.py --config cfg5shard --e2e --no-cpu-baseline --headline-only 2> $O/bench_cfg5_e2e.err | grep "^{" > $O/cfg5shard_e2e_bench.json; echo "cfg5 e2e rc=$?"
python - <<'PY'
import json
for f in ("cfg2_bench.json", "cfg5shard_e2e_bench.json"):
    try:
        d = json.loads(open("gpurun_out/r05final/" + f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f, d["value"], d["ms_per_step"], r["frac"], r.get("traffic_over_algorithmic"), d.get("self_check", {}).get("matches_pinned"), json.dumps(d.get("e2e_ranks"))[:500])
        for k, v in (d.get("configs") or {}).items():
            rr = v.get("roofline") or {}
            print("   ", k, v.get("value"), v.get("ms_per_step"), rr.get("frac"), rr.get("useful_frac"), rr.get("traffic_over_algorithmic") or (rr.get("hbm") or {}).get("traffic_over_algorithmic"), (v.get("self_check") or {}).get("matches_pinned"), v.get("error"))
        e = d.get("e2e") or {}
        print("    e2e", {k: (v.get("mreads_s") if isinstance(v, dict) else v) for k, v in e.items() if k not in ("sharded_big", "default_invocation")}, {k: (v.get("mreads_s") if isinstance(v, dict) else v) for k, v in e.get("sharded_big", {}).items()})
    except Exception as ex:
        print(f, "unreadable", ex)
PY
