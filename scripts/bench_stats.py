#!/usr/bin/env python3
"""Kernel time of fxg_run_quality_stats on the cfg2-sized batch (50 M x 150 bp), not a test."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fastx_toolkit_amd import Engine
eng = Engine(0)
L = int(os.environ.get("LEN", "150"))
n = int(os.environ.get("READS", str(50000000 * 150 // L)))
b, q = eng.synth(2, 0, n, L, False)
eng.set_profiling(True)
ms = []
for _ in range(4):
    h = eng.quality_stats(b, q, fixed_len=L, sync=False)
    ms.append(eng.last_kernel_ms())
eng.sync()
li = eng.last_launch()
print(json.dumps(dict(kernel=li["kernel"], grid=li["grid"], ms_min=round(min(ms), 3), GBs=round(2 * n * L / min(ms) / 1e6, 1), frac_hbm=round(2 * n * L / min(ms) / 1e6 / 8000, 3),
                      total=int(h.sum()) // 4, expect=n * L)))
