#!/bin/bash
# round 6, call n: short reads through fxg_kernel_rows_multi (several reads per lane) against the tile kernel; the quality-kernel parity tests
O=gpurun_out/r06n; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "quality_kernels or rows_kernel or fuzz or config or variable" > $O/pytest_rows.txt 2>&1; tail -n 6 $O/pytest_rows.txt
timeout 900 python scripts/rows_vs_tiles_speed.py 28 32 36 40 44 50 56 60 72 76 79 80 100 150 > $O/rows_multi_short_reads.txt 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r06n/rows_multi_short_reads.txt"):
    if l.startswith("{"):
        d = json.loads(l); print("L %3d %-30s ms %7.3f  %5.2f TB/s algorithmic = %.3f of 8 TB/s  kept %d" % (d["read_len"], d["kernel"], d["ms_min"], d["alg_TBs"], d["alg_TBs"] / 8.0, d["kept"]))
    else: print(l.strip()[:200])
PY
