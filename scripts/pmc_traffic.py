#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE of the cfg2 kernel -> profiles/pmc_traffic.json (+ the raw CSVs under profiles/<tag>_pmc/).

On the GPU box (separate passes, counters only -- MI355X_MICROARCH.md, HBM section):
    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc/FETCH_SIZE -o pmc --output-format csv -- python $R/scripts/pmc_run.py
    rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/pmc/WRITE_SIZE -o pmc --output-format csv -- python $R/scripts/pmc_run.py
then here:  python scripts/pmc_traffic.py gpurun_out/pmc r01b
The 1 GiB clone in pmc_run.py calibrates both counters on this rocprofv3 (gfx950: FETCH_SIZE reports 1/2 of coalesced reads).
"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from pmc_parse import load  # noqa: E402

src, tag = sys.argv[1], sys.argv[2]
R = int(os.environ.get("READS", "50000000"))
out = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    d = load(os.path.join(src, ctr))
    tiles = [k for k in d if "fxg_kernel_rows" in k] or [k for k in d if "fxg_kernel_tiles" in k]   # the kernel cfg2 ran as
    clone = [k for k in d if "elementwise" in k.lower() or "copy" in k.lower()]
    assert tiles, list(d)
    v = d[tiles[0]][ctr]
    out[ctr] = dict(kernel=tiles[0], per_launch_kb=sum(v) / len(v), n=len(v))
    cal = None
    for k in clone:                                   # the 1 GiB clone: the launch whose counter is closest to 2^20 KB (or half of it)
        for x in d[k][ctr]:
            if 0.4 * 2 ** 20 < x < 1.2 * 2 ** 20:
                cal = x / 2 ** 20
    out[ctr]["calibration_fraction"] = cal
    dst = os.path.join(ROOT, "profiles", tag + "_pmc")
    os.makedirs(dst, exist_ok=True)
    shutil.copy(os.path.join(src, ctr, "pmc_counter_collection.csv"), os.path.join(dst, ctr.lower() + "_counter_collection.csv"))
fcal = out["FETCH_SIZE"]["calibration_fraction"] or 0.5
wcal = out["WRITE_SIZE"]["calibration_fraction"] or 1.0
rd = out["FETCH_SIZE"]["per_launch_kb"] * 1024 / fcal
wr = out["WRITE_SIZE"]["per_launch_kb"] * 1024 / wcal
kept_bytes = int(os.environ.get("KEPT_BYTES", "3558930905"))
j = {
    "kernel": out["FETCH_SIZE"]["kernel"] + " (cfg2, %d x 150bp, compaction)" % R,
    "reads_per_launch": R,
    "FETCH_SIZE_KB_per_launch": round(out["FETCH_SIZE"]["per_launch_kb"], 1),
    "WRITE_SIZE_KB_per_launch": round(out["WRITE_SIZE"]["per_launch_kb"], 1),
    "calibration": {"workload": "1 GiB torch clone in the same rocprofv3 run", "FETCH_SIZE_reported_fraction": round(fcal, 4),
                    "WRITE_SIZE_reported_fraction": round(wcal, 4),
                    "note": "counters divided by the fraction the 1 GiB clone reports" + ("" if "rows" in out["FETCH_SIZE"]["kernel"] else "; the read correction is applied to the unaligned 16 B window loads of the gather as well (uncalibrated for that pattern)") + "; Infinity-Cache hits are counted"},
    "hbm_read_bytes_per_launch": int(rd), "hbm_write_bytes_per_launch": int(wr), "hbm_bytes_per_launch": int(rd + wr),
    "algorithmic_bytes_per_launch": R * 304 + 2 * kept_bytes,
    "tag": tag,
    "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python scripts/pmc_run.py",
    "kernel_version": os.environ.get("KERNEL_VERSION", tag),
}
json.dump(j, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(j, indent=1))
