#!/usr/bin/env python3
"""Cycles per row of the clip DP as its waves see them (GPU box, -DFXG_ABL_ROWCLK -DFXG_DBG_BITS build; not a test)."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fastx_toolkit_amd import Engine, make_params
AD = b"AGATCGGAAGAGC"
eng = Engine(0)
for cfg in os.environ.get("CFGS", "cfg3,cfg5").split(","):
    R, L = int(os.environ.get("READS", "20000000")), (100 if cfg == "cfg3" else 150)
    b, q = eng.synth(3 if cfg == "cfg3" else 5, 0, R, L, True)
    P = (make_params(stages=1, adapter=AD, clip_min_len=15, clip_flags=4) if cfg == "cfg3" else
         make_params(stages=7, adapter=AD, clip_min_len=15, clip_flags=4, qt_threshold=20, qt_min_len=30, qf_min_quality=20, qf_min_percent=80))
    for compact in (True, False):
        outs = eng.alloc_outputs(R, L, compact=compact, meta=False)
        eng.set_profiling(True)
        for _ in range(2):
            r = eng.run(b, q, P, fixed_len=L, compact=compact, meta=False, outputs=outs)
        ms = eng.last_kernel_ms()
        ph = (ctypes.c_uint64 * 11)()
        eng.lib.fxg_debug_phase_clocks(eng.ctx, ph)
        li = eng.last_launch()
        waves = (R + 63) // 64
        print(json.dumps(dict(cfg=cfg, compact=compact, debug=os.environ.get("FXG_DEBUG", "0"), ms=round(ms, 3), grid=li["grid"], pass1_cycles_per_wave_row=round(ph[0] / max(1, ph[1]), 1), pass1_rows_per_wave=round(ph[1] / waves, 1),
                              pass2_cycles_per_wave_row=round(ph[2] / max(1, ph[3]), 1), pass2_rows_per_wave=round(ph[3] / waves, 2), pass2_score_only_rows_per_wave=round(ph[4] / waves, 2),
                              pass1_ms_per_wave_share=round(ph[0] / 2.4e6 / li["grid"] / 4, 3), pass2_ms_per_wave_share=round(ph[2] / 2.4e6 / li["grid"] / 4, 3))), flush=True)
        del outs
eng.close()
