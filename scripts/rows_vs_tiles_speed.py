#!/usr/bin/env python3
"""GPU box: kernel time of the quality trim+filter pass with fxg_kernel_rows and with fxg_kernel_tiles<0,0> for several read lengths
(about 7.5 GB of bases per run): where each kernel pays.  One JSON line per (length, kernel)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fastx_toolkit_amd import Engine, make_params  # noqa: E402

eng = Engine(0)
eng.set_profiling(True)
P = make_params(stages=6, qt_threshold=20, qt_min_len=10, qf_min_quality=20, qf_min_percent=80)
for L in [int(x) for x in (sys.argv[1:] or ["36", "50", "76", "100", "125", "150", "200", "250", "300"])]:
    n = min(7_500_000_000 // L, 200_000_000)
    b, q = eng.synth(2, 0, n, L, False, L)
    outs = eng.alloc_outputs(n, L, compact=True, meta=False)
    for rows in ("2", "0"):                  # 2: the register-row kernel wherever it exists (to 304 bytes), 0: the tile kernel
        os.environ["FXG_ROWS"] = rows
        ms = []
        for _ in range(5):
            r = eng.run(b, q, P, fixed_len=L, compact=True, meta=False, outputs=outs)
            ms.append(eng.last_kernel_ms())
        li = eng.last_launch()
        c = r.counters
        alg = n * (2 * L + 4) + 2 * int(c[2])
        print(json.dumps(dict(read_len=L, reads=n, kernel=li["kernel"].split(" ")[0], ms_min=round(min(ms[1:]), 3), ms_avg=round(sum(ms[1:]) / 4, 3),
                              greads_s=round(n / min(ms[1:]) / 1e6, 2), alg_TBs=round(alg / min(ms[1:]) / 1e9, 2), kept=int(c[1]))), flush=True)
    del b, q, outs
