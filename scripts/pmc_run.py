#!/usr/bin/env python3
"""Workload for the PMC passes: a known-size device copy (calibration of FETCH_SIZE / WRITE_SIZE on this
rocprofv3) followed by the cfg2 pipeline at bench size.  Run under
    rocprofv3 --pmc FETCH_SIZE -d <dir> -o fetch --output-format csv -- python scripts/pmc_run.py
    rocprofv3 --pmc WRITE_SIZE -d <dir> -o write --output-format csv -- python scripts/pmc_run.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from fastx_toolkit_amd import Engine, make_params  # noqa: E402

R = int(os.environ.get("READS", "50000000"))
eng = Engine(0)
x = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")   # 1 GiB
x.random_(0, 255)
for _ in range(3):
    y = x.clone()          # calibration: reads 2^30 bytes, writes 2^30 bytes
torch.cuda.synchronize()
b, q = eng.synth(2, 0, R, 150)
P = make_params(stages=6, qt_threshold=20, qt_min_len=30, qf_min_quality=20, qf_min_percent=80)
outs = eng.alloc_outputs(R, 150, compact=True, meta=False)
for _ in range(4):
    r = eng.run(b, q, P, fixed_len=150, compact=True, meta=False, outputs=outs)
c = r.counters
print("kept", int(c[1]), "kept_bytes", int(c[2]), "alg_bytes", R * 304 + 2 * int(c[2]))
if os.environ.get("ALSO_DECISION"):
    for _ in range(3):
        eng.run(b, q, P, fixed_len=150, compact=False)
    eng.sync()
