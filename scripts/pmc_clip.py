#!/usr/bin/env python3
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fastx_toolkit_amd import Engine, make_params
eng = Engine(0)
n, L = 20_000_000, 100
b, q = eng.synth(3, 0, n, L, True)
P = make_params(stages=1, adapter=b"AGATCGGAAGAGC", clip_min_len=15, clip_flags=4)
outs = eng.alloc_outputs(n, L, compact=True, meta=False)
for _ in range(2):
    r = eng.run(b, q, P, fixed_len=L, compact=True, meta=False, outputs=outs)
print(int(r.counters[1]), "cells", n * L * 13)
