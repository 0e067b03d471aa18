#!/bin/bash
# round 6, call aj: HEAD against this session's starting point (fa857ca built as libfxg_v_r6start.so) in ONE call, alternating: statistics kernel (8 rounds), cfg2 (4),
# cfg3 / cfg5 at 20 M reads (2 rounds of 7 launches)
O=gpurun_out/r06aj; mkdir -p $O
for rep in 1 2 3 4 5 6 7 8; do for v in libfxg_v_r6start.so libfxg.so; do
  echo -n "$v: "; FXG_LIB=$PWD/fastx_toolkit_amd/$v timeout 300 python scripts/bench_stats.py 2>&1 | tail -n 1 | cut -c1-110
done; done > $O/stats_head_vs_start.txt 2>&1
for rep in 1 2 3 4; do for v in libfxg_v_r6start.so libfxg.so; do
  echo -n "$v: "; ONLY=cfg2 FXG_LIB=$PWD/fastx_toolkit_amd/$v timeout 300 python scripts/bench_configs.py 2>&1 | tail -n 1 | cut -c1-160
done; done > $O/cfg2_head_vs_start.txt 2>&1
L=fastx_toolkit_amd/libfxg_v_r6start.so,fastx_toolkit_amd/libfxg.so
LIBS=$L,$L CFGS=cfg3,cfg5 READS=20000000 REPS=7 timeout 900 python scripts/clip_ab.py > $O/clip_head_vs_start.txt 2>&1
python - <<'PY'
import json, re, collections
for f in ("stats_head_vs_start", "cfg2_head_vs_start"):
    d = collections.defaultdict(list)
    for l in open("gpurun_out/r06aj/%s.txt" % f):
        m = re.match(r"(\S+): .*\"ms_min\": ([0-9.]+)", l)
        if m: d[m.group(1)].append(float(m.group(2)))
    for k, v in d.items(): print(f, k, "mean %.3f" % (sum(v) / len(v)), sorted(v))
for l in open("gpurun_out/r06aj/clip_head_vs_start.txt"):
    if l.startswith("{"):
        x = json.loads(l); print(x["lib"].split("/")[-1], x["cfg"], x["ms_min"], x["ms_med"], x["checksum"])
PY
