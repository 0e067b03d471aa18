#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE of a config's dominant kernel -> profiles/pmc_traffic_<cfg>.json (+ the raw CSVs under profiles/<tag>_pmc_<cfg>/).

    python scripts/pmc_traffic.py <dir with FETCH_SIZE/ and WRITE_SIZE/ passes> <tag> <cfg>
The 1 GiB clone in pmc_run.py calibrates both counters on this rocprofv3 (gfx950: FETCH_SIZE reports 1/2 of coalesced reads).
The JSON records the hash of the kernel sources it was collected on; bench.py attaches it only to lines built from the same sources.
"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
sys.path.insert(0, ROOT)
from pmc_parse import load  # noqa: E402
import bench  # noqa: E402

src, tag, cfgname = sys.argv[1], sys.argv[2], sys.argv[3]
cfg = bench.CONFIGS[cfgname]
R = int(os.environ.get("READS", str(cfg["reads"])))
out = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    d = load(os.path.join(src, ctr))
    mine = [k for k in d if "fxg_kernel_rows" in k or "fxg_kernel_tiles" in k or ("fxg_kernel_quality_stats" in k and "fold" not in k)]
    clone = [k for k in d if "elementwise" in k.lower() or "copy" in k.lower()]
    assert mine, list(d)
    k0 = max(mine, key=lambda k: sum(d[k][ctr]))            # the dominant kernel of the run
    v = d[k0][ctr]
    out[ctr] = dict(kernel=k0, per_launch_kb=sum(v) / len(v), n=len(v))
    cal = None
    for k in clone:                                   # the 1 GiB clone: the launch whose counter is closest to 2^20 KB (or half of it)
        for x in d[k][ctr]:
            if 0.4 * 2 ** 20 < x < 1.2 * 2 ** 20:
                cal = x / 2 ** 20
    out[ctr]["calibration_fraction"] = cal
    dst = os.path.join(ROOT, "profiles", "%s_pmc_%s" % (tag, cfgname))
    os.makedirs(dst, exist_ok=True)
    shutil.copy(os.path.join(src, ctr, "pmc_counter_collection.csv"), os.path.join(dst, ctr.lower() + "_counter_collection.csv"))
fcal = out["FETCH_SIZE"]["calibration_fraction"] or 0.5
wcal = out["WRITE_SIZE"]["calibration_fraction"] or 1.0
rd = out["FETCH_SIZE"]["per_launch_kb"] * 1024 / fcal
wr = out["WRITE_SIZE"]["per_launch_kb"] * 1024 / wcal
exp = (bench.EXPECTED.get(cfgname) or [None])[0]            # (one tuple per rank since round 6: the traffic is collected on the shard that starts at read 0)
alg = R * 2 * cfg["L"] if cfg["params"] is None else R * (2 * cfg["L"] + 4) + 2 * int(os.environ.get("KEPT_BYTES", exp[1] if exp and R == cfg["reads"] else 0))
kname = out["FETCH_SIZE"]["kernel"].replace("void ", "").replace(" ", "")
j = {
    "config": cfgname, "kernel": out["FETCH_SIZE"]["kernel"] + " (%s, %d x %dbp)" % (cfgname, R, cfg["L"]), "kernel_name": kname.split("(")[0],
    "reads_per_launch": R,
    "FETCH_SIZE_KB_per_launch": round(out["FETCH_SIZE"]["per_launch_kb"], 1),
    "WRITE_SIZE_KB_per_launch": round(out["WRITE_SIZE"]["per_launch_kb"], 1),
    "calibration": {"workload": "1 GiB torch clone in the same rocprofv3 run", "FETCH_SIZE_reported_fraction": round(fcal, 4),
                    "WRITE_SIZE_reported_fraction": round(wcal, 4),
                    "note": "counters divided by the fraction the 1 GiB clone reports (the read correction is calibrated on coalesced 16 B/lane streaming loads; "
                            "unaligned 16 B windows of a gather are uncalibrated); Infinity-Cache hits are counted"},
    "hbm_read_bytes_per_launch": int(rd), "hbm_write_bytes_per_launch": int(wr), "hbm_bytes_per_launch": int(rd + wr),
    "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": round((rd + wr) / alg, 4) if alg else None,
    "tag": tag, "csrc_sha16": (open(os.path.join(src, "csrc_sha16.txt")).read().strip() if os.path.exists(os.path.join(src, "csrc_sha16.txt")) else bench.csrc_sha16()),
    "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- CFG=%s python scripts/pmc_run.py" % cfgname,
}
json.dump(j, open(os.path.join(ROOT, "profiles", "pmc_traffic_%s.json" % cfgname), "w"), indent=1)
print(json.dumps(j, indent=1))
