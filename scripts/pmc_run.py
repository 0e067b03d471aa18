#!/usr/bin/env python3
"""Workload for the PMC passes: a known-size device copy (calibration of FETCH_SIZE / WRITE_SIZE on this rocprofv3) followed by one
BASELINE config at bench size (CFG = cfg2 | cfg3 | cfg4 | cfg5shard | stats, bench.py's CONFIGS).  Run under
    rocprofv3 --pmc FETCH_SIZE -d <dir>/FETCH_SIZE -o pmc --output-format csv -- python scripts/pmc_run.py
    rocprofv3 --pmc WRITE_SIZE -d <dir>/WRITE_SIZE -o pmc --output-format csv -- python scripts/pmc_run.py
(separate passes, counters only -- MI355X_MICROARCH.md, HBM section), then scripts/pmc_traffic.py turns the CSVs into
profiles/pmc_traffic_<cfg>.json.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from fastx_toolkit_amd import Engine, make_params  # noqa: E402

CFG = os.environ.get("CFG", "cfg2")
cfg = bench.CONFIGS[CFG]
R = int(os.environ.get("READS", str(cfg["reads"])))
L = cfg["L"]
eng = Engine(0)
x = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")   # 1 GiB
x.random_(0, 255)
for _ in range(3):
    y = x.clone()          # calibration: reads 2^30 bytes, writes 2^30 bytes
torch.cuda.synchronize()
del x, y
b, q = eng.synth(cfg["seed"], 0, R, L, cfg["adapter"])
if cfg["params"] is None:
    hist = torch.zeros((L, 5, 128), dtype=torch.int64, device=eng.device)
    for _ in range(4):
        eng.quality_stats(b, q, fixed_len=L, hist=hist, sync=False)
    eng.sync()
    print("cfg", CFG, "reads", R, "alg_bytes", R * 2 * L)
else:
    P = make_params(**cfg["params"])
    outs = eng.alloc_outputs(R, L, compact=True, meta=False)
    for _ in range(4):
        r = eng.run(b, q, P, fixed_len=L, compact=True, meta=False, outputs=outs)
    c = r.counters
    print("cfg", CFG, "reads", R, "kept", int(c[1]), "kept_bytes", int(c[2]), "alg_bytes", R * (2 * L + 4) + 2 * int(c[2]), "kernel", eng.last_launch()["kernel"])
