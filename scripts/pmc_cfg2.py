#!/usr/bin/env python3
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fastx_toolkit_amd import Engine, make_params
eng = Engine(0)
R = 50_000_000
b, q = eng.synth(2, 0, R, 150)
P = make_params(stages=6, qt_threshold=20, qt_min_len=30, qf_min_quality=20, qf_min_percent=80)
outs = eng.alloc_outputs(R, 150, compact=True, meta=False)
for _ in range(3):
    r = eng.run(b, q, P, fixed_len=150, compact=True, meta=False, outputs=outs)
print(int(r.counters[1]))
