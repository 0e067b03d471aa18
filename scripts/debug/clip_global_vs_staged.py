#!/usr/bin/env python3
"""GPU box: the register two-pass clip instances with the DP over the staged tile (FXG_CLIP_GLOBAL=0) and straight over the batch (1), per read length."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fastx_toolkit_amd import Engine, make_params
eng = Engine(0)
AD = os.environ.get("ADAPTER", "AGATCGGAAGAGC").encode()
SHAPES = ((100, 20_000_000, 1), (150, 20_000_000, 7), (152, 10_000_000, 1), (200, 10_000_000, 1), (252, 8_000_000, 1), (300, 6_000_000, 1), (300, 6_000_000, 7), (1000, 2_000_000, 1))
if os.environ.get('SHAPES'):
    SHAPES = tuple(tuple(int(x) for x in t.split(':')) for t in os.environ['SHAPES'].split(','))
for L, R, stages in SHAPES:
    b, q = eng.synth(5, 0, R, L, True)
    P = (make_params(stages=1, adapter=AD, clip_min_len=15, clip_flags=4) if stages == 1 else
         make_params(stages=7, adapter=AD, clip_min_len=15, clip_flags=4, qt_threshold=20, qt_min_len=30, qf_min_quality=20, qf_min_percent=80))
    outs = eng.alloc_outputs(R, L, compact=True, meta=False)
    eng.set_profiling(True)
    row = dict(L=L, reads=R, stages=stages)
    for mode in ("0", "1", None):
        if mode is None: os.environ.pop("FXG_CLIP_GLOBAL", None)
        else: os.environ["FXG_CLIP_GLOBAL"] = mode
        ms = []
        for _ in range(3):
            r = eng.run(b, q, P, fixed_len=L, compact=True, meta=False, outputs=outs); ms.append(eng.last_kernel_ms())
        li = eng.last_launch()
        row["default" if mode is None else ("global" if mode == "1" else "staged")] = dict(ms=round(min(ms), 3), gcups=round(R * L * len(AD) / min(ms) / 1e6, 0), kernel=li['kernel'].split()[0], tile=li["tile_reads"], lds=li["lds"], kept=int(r.counters[1]))
    print(json.dumps(row), flush=True)
    del b, q, outs
