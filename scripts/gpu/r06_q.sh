#!/bin/bash
# round 6, call q: the buckets of 44, 52, 60, 72 and 88 columns: parity tests and every adapter length again
O=gpurun_out/r06q; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "clip or fuzz" > $O/pytest_clip.txt 2>&1; tail -n 4 $O/pytest_clip.txt
READS=10000000 timeout 900 python scripts/clip_by_adapter_len.py 13 16 17 20 21 24 28 32 33 36 37 40 41 44 45 48 49 52 53 56 57 60 61 64 65 72 73 80 81 88 89 99 > $O/clip_by_adapter_len.txt 2>&1
READS=10000000 L=150 timeout 900 python scripts/clip_by_adapter_len.py 13 34 44 52 60 64 72 88 99 > $O/clip_by_adapter_len_150.txt 2>&1
python - <<'PY'
import json
for f in ("clip_by_adapter_len", "clip_by_adapter_len_150"):
    print(f); prev = None
    for l in open("gpurun_out/r06q/%s.txt" % f):
        if l.startswith("{"):
            d = json.loads(l); step = "" if prev is None else "  step %+.0f %%" % (100.0 * (d["ms_min"] / prev - 1)); prev = d["ms_min"]
            print("  A %3d %-28s ms %7.3f  gcups %7.1f%s" % (d["adapter_len"], d["kernel"].split(" ")[0], d["ms_min"], d["gcups"], step))
PY
