#!/bin/bash
# round 6, call l: every adapter length through the pair table (17..99 columns: fxg_clip_row_kt), with and without N; the clip parity tests
O=gpurun_out/r06l; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "clip or fuzz or config or cfg5 or cfg3 or long_reads" > $O/pytest_clip.txt 2>&1; tail -n 6 $O/pytest_clip.txt
READS=10000000 timeout 900 python scripts/clip_by_adapter_len.py 8 13 16 17 20 24 28 32 33 34 36 40 44 48 49 56 57 64 65 72 80 81 99 > $O/clip_by_adapter_len.txt 2>&1
WITH_N=1 READS=10000000 timeout 900 python scripts/clip_by_adapter_len.py 13 16 24 34 40 48 56 64 > $O/clip_by_adapter_len_with_n.txt 2>&1
READS=10000000 L=150 timeout 900 python scripts/clip_by_adapter_len.py 13 34 48 64 99 > $O/clip_by_adapter_len_150.txt 2>&1
python - <<'PY'
import json
for f in ("clip_by_adapter_len", "clip_by_adapter_len_with_n", "clip_by_adapter_len_150"):
    print(f)
    for l in open("gpurun_out/r06l/%s.txt" % f):
        if l.startswith("{"):
            d = json.loads(l); print("  A %3d %-28s ms %7.3f  gcups %7.1f  kept %d %d" % (d["adapter_len"], d["kernel"].split(" ")[0], d["ms_min"], d["gcups"], d["kept"], d["kept_bases"]))
        else: print("  ", l.strip()[:200])
PY
