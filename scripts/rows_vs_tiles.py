#!/usr/bin/env python3
"""GPU box: fxg_kernel_rows against fxg_kernel_tiles<0,0> on the same batches (every output array), first difference printed."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fastx_toolkit_amd import Engine, make_params  # noqa: E402

eng = Engine(0)
bad = 0
for seed, n, L, stride, var in [(1, 64, 150, 150, False), (2, 100, 150, 150, False), (3, 5000, 150, 150, False), (4, 3000, 36, 36, False), (5, 2000, 100, 100, False),
                                (6, 4000, 150, 152, True), (7, 1000, 13, 29, True), (8, 333, 7, 7, False), (9, 100000, 150, 150, False), (10, 2500, 101, 104, True)]:
    import torch
    b, q = eng.synth(seed, 0, n, L, False, stride)
    lens = None
    if var:
        rng = np.random.default_rng(seed)
        lens = torch.from_numpy(rng.integers(1, L + 1, n).astype(np.int16)).to(eng.device)
    for pd in (dict(stages=6, qt_threshold=20, qt_min_len=30, qf_min_quality=20, qf_min_percent=80), dict(stages=2, qt_threshold=25, qt_min_len=1),
               dict(stages=4, qf_min_quality=15, qf_min_percent=50), dict(stages=6, qt_threshold=30, qt_min_len=2, qf_min_quality=10, qf_min_percent=10)):
        P = make_params(**pd)
        res = {}
        for rows in ("1", "0"):
            os.environ["FXG_ROWS"] = rows
            r = eng.run(b, q, P, lens=lens, fixed_len=None if var else L, compact=True, meta=True)
            res[rows] = (eng.last_launch()["kernel"], r.to_host())
        ka, a = res["1"]
        kb, c = res["0"]
        for key in c:
            x, y = np.asarray(a[key]), np.asarray(c[key])
            if key in ("out_bases", "out_qual"):
                nb = int(c["counters"][2]); x, y = x.reshape(-1)[:nb], y.reshape(-1)[:nb]
            elif key in ("out_len", "kept_index", "out_off"):
                nk = int(c["counters"][1]); x, y = x[:nk], y[:nk]
            if x.shape != y.shape or not np.array_equal(x, y):
                d = np.flatnonzero(x.reshape(-1) != y.reshape(-1)) if x.shape == y.shape else []
                print("DIFF", (seed, n, L, stride, var), pd, key, ka, "vs", kb, "first", d[:8], "count", len(d))
                bad += 1
                break
print("rows_vs_tiles:", "OK" if not bad else "%d differences" % bad)
