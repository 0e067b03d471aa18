#!/bin/bash
# round 6, call ak: the statistics kernel's variants in ONE call, eight alternating rounds: the session's starting point, and HEAD's loop by trip order (1 round robin,
# 2 contiguous slices), depth (1, 3) and load policy (nt, default)
O=gpurun_out/r06ak; mkdir -p $O
for rep in 1 2 3 4 5 6 7 8; do
for cfg in "libfxg_v_r6start.so 1" "libfxg.so 1" "libfxg.so 2" "libfxg_v_qsd3nontl.so 1" "libfxg_v_qsd3nontl.so 2" "libfxg_v_qsd1.so 1" "libfxg_v_qsd1.so 2" "libfxg_v_qsd1nontl.so 1" "libfxg_v_qsd1nontl.so 2"; do
  set -- $cfg
  echo -n "$1 order=$2: "; FXG_QS_ROUND_ROBIN=$2 FXG_LIB=$PWD/fastx_toolkit_amd/$1 timeout 300 python scripts/bench_stats.py 2>&1 | tail -n 1 | cut -c1-110
done; done > $O/stats_variants.txt 2>&1
python - <<'PY'
import re, collections
d = collections.defaultdict(list)
for l in open("gpurun_out/r06ak/stats_variants.txt"):
    m = re.match(r"(\S+ order=\d): .*\"ms_min\": ([0-9.]+)", l)
    if m: d[m.group(1)].append(float(m.group(2)))
for k, v in d.items(): print("%-36s mean %.3f  median %.3f  %s" % (k, sum(v) / len(v), sorted(v)[len(v) // 2], sorted(v)))
PY
