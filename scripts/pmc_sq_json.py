#!/usr/bin/env python3
"""SQ_INSTS_VALU of a clip config's dominant kernel -> profiles/pmc_sq_<cfg>.json (what the kernel ISSUES per cell of the L x 13 matrix).

    python scripts/pmc_sq_json.py <dir of one `rocprofv3 --pmc SQ_INSTS_VALU -- CFG=<cfg> python scripts/pmc_run.py` pass> <tag> <cfg>
bench.py attaches the figure (roofline.issued_*) only to lines built from the same kernel sources (csrc_sha16).
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
d = load(src)
mine = [k for k in d if "fxg_kernel_tiles" in k]
assert mine, list(d)
k0 = max(mine, key=lambda k: sum(d[k]["SQ_INSTS_VALU"]))
v = d[k0]["SQ_INSTS_VALU"]
cells = R * cfg["L"] * len(bench.ADAPTER)
per_launch = sum(v) / len(v)
dst = os.path.join(ROOT, "profiles", "%s_pmc_sq_%s" % (tag, cfgname))
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "pmc_counter_collection.csv"), os.path.join(dst, "sq_insts_valu_counter_collection.csv"))
sha = open(os.path.join(src, "..", "csrc_sha16.txt")).read().strip() if os.path.exists(os.path.join(src, "..", "csrc_sha16.txt")) else bench.csrc_sha16()
j = {"config": cfgname, "kernel": k0, "reads_per_launch": R, "cells_per_launch": cells, "SQ_INSTS_VALU_per_launch": per_launch, "launches": len(v),
     "valu_instr_per_cell": round(per_launch * 64.0 / cells, 3), "tag": tag, "csrc_sha16": sha,
     "command": "rocprofv3 --pmc SQ_INSTS_VALU -- CFG=%s python scripts/pmc_run.py" % cfgname,
     "note": "wave-instructions x 64 lanes / cells of the L x 13 matrix the reference fills: everything the kernel issues (both passes, staging, scan, write-out)"}
json.dump(j, open(os.path.join(ROOT, "profiles", "pmc_sq_%s.json" % cfgname), "w"), indent=1)
print(json.dumps(j))
