#!/usr/bin/env python3
"""Workload of the SQ counter passes of the clip kernel (scripts/pmc_sq.sh): CFG = cfg3 (20 M x 100) | cfg5 (20 M x 150, clip + trim + filter)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fastx_toolkit_amd import Engine, make_params
eng = Engine(0)
cfg5 = os.environ.get("CFG", "cfg3") == "cfg5"
n, L = 20_000_000, (150 if cfg5 else 100)
b, q = eng.synth(5 if cfg5 else 3, 0, n, L, True)
AD = b"AGATCGGAAGAGC"
P = (make_params(stages=7, adapter=AD, clip_min_len=15, clip_flags=4, qt_threshold=20, qt_min_len=30, qf_min_quality=20, qf_min_percent=80) if cfg5
     else make_params(stages=1, adapter=AD, clip_min_len=15, clip_flags=4))
outs = eng.alloc_outputs(n, L, compact=True, meta=False)
for _ in range(2):
    r = eng.run(b, q, P, fixed_len=L, compact=True, meta=False, outputs=outs)
print(int(r.counters[1]), "reads", n, "cells", n * L * 13)
