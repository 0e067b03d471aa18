#!/bin/bash
# round 6, call m: two passes wherever the summary rows are fewer than the read's rows (44..72 columns on 100-base reads); clip parity tests again
O=gpurun_out/r06m; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "clip or fuzz or config or cfg5 or cfg3 or long_reads" > $O/pytest_clip.txt 2>&1; tail -n 4 $O/pytest_clip.txt
READS=10000000 timeout 900 python scripts/clip_by_adapter_len.py 17 20 33 34 40 44 48 49 56 57 64 65 72 73 80 81 99 > $O/clip_by_adapter_len.txt 2>&1
WITH_N=1 READS=10000000 timeout 900 python scripts/clip_by_adapter_len.py 48 56 64 > $O/clip_by_adapter_len_with_n.txt 2>&1
python - <<'PY'
import json
for f in ("clip_by_adapter_len", "clip_by_adapter_len_with_n"):
    print(f)
    for l in open("gpurun_out/r06m/%s.txt" % f):
        if l.startswith("{"):
            d = json.loads(l); print("  A %3d %-28s ms %7.3f  gcups %7.1f  kept %d %d" % (d["adapter_len"], d["kernel"].split(" ")[0], d["ms_min"], d["gcups"], d["kept"], d["kept_bases"]))
PY
