#!/usr/bin/env python3
"""Kernel-level timing of every BASELINE.json config on one GPU (not the driver's bench; feeds DESIGN.md)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fastx_toolkit_amd import Engine, make_params  # noqa: E402

AD = b"AGATCGGAAGAGC"
SCALE = float(os.environ.get("SCALE", "1.0"))
CFGS = [
    ("cfg2 qtrim+qfilter 50M x150", (2, 50_000_000, 150, False), dict(stages=6, qt_threshold=20, qt_min_len=30, qf_min_quality=20, qf_min_percent=80), 0),
    ("cfg3 clip 50M x100", (3, 50_000_000, 100, True), dict(stages=1, adapter=AD, clip_min_len=15, clip_flags=4), 13),
    ("cfg4 revcomp+ftrim 200M x150", (2, 200_000_000, 150, False), dict(stages=24, ft_first=5, ft_last=145), 0),
    ("cfg5 clip+qtrim+qfilter 125M x150 (1/8 of 1B)", (5, 125_000_000, 150, True),
     dict(stages=7, adapter=AD, clip_min_len=15, clip_flags=4, qt_threshold=20, qt_min_len=30, qf_min_quality=20, qf_min_percent=80), 13),
]
only = os.environ.get("ONLY")
eng = Engine(0)
eng.set_profiling(True)
for name, (seed, n, L, ad), pd, alen in CFGS:
    if only and only not in name:
        continue
    n = int(n * SCALE)
    b, q = eng.synth(seed, 0, n, L, ad)
    outs = eng.alloc_outputs(n, L, compact=True, meta=False)
    P = make_params(**pd)
    ms = []
    for _ in range(4):
        r = eng.run(b, q, P, fixed_len=L, compact=True, meta=False, outputs=outs)
        ms.append(eng.last_kernel_ms())
    c = r.counters
    kept, kb = int(c[1]), int(c[2])
    alg = n * (2 * L + 4) + 2 * kb
    t = min(ms) * 1e-3
    print(json.dumps(dict(config=name, kernel=eng.last_launch()["kernel"], grid=eng.last_launch()["grid"], ms_min=round(min(ms), 3),
                          ms_avg=round(sum(ms) / len(ms), 3), mreads_s=round(n / t / 1e6, 1), gbases_s=round(n * L / t / 1e9, 1),
                          alg_GBs=round(alg / t / 1e9, 1), frac_hbm=round(alg / t / 8e12, 3),
                          gcups=round(n * L * alen / t / 1e9, 1) if alen else None, kept=kept, kept_bases=kb)), flush=True)
    del b, q, outs, r
    import torch
    torch.cuda.empty_cache()
