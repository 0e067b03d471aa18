#!/usr/bin/env python3
"""Timing ablations of the cfg2 kernel on the GPU box (not a test): FXG_DEBUG bits switch phases off."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import subprocess  # noqa: E402
from fastx_toolkit_amd import build as _b  # noqa: E402

if not os.environ.get("FXG_LIB"):                  # FXG_LIB preset: time that (prebuilt variant) library as is
    ABL = os.path.join(_b.PKG, "libfxg_ablation.so")   # separate build: the product library has no ablation switches
    subprocess.check_call([_b.hipcc()] + _b.HIPCC_FLAGS + ["-DFXG_ABLATION", os.path.join(_b.CSRC, "fxg_engine.hip"), "-o", ABL])
    os.environ["FXG_LIB"] = ABL
from fastx_toolkit_amd import Engine, make_params  # noqa: E402

CFG = os.environ.get("CFG", "cfg2")
R = int(os.environ.get("READS", "200000000" if CFG == "cfg4" else "50000000"))
eng = Engine(0)
b, q = eng.synth(2, 0, R, 150)
P = (make_params(stages=24, ft_first=5, ft_last=145) if CFG == "cfg4"
     else make_params(stages=6, qt_threshold=20, qt_min_len=30, qf_min_quality=20, qf_min_percent=80))
outs = eng.alloc_outputs(R, 150, compact=True, meta=False)
eng.set_profiling(True)


def t(label, env, compact=True, reps=4):
    for k in ("FXG_DEBUG", "FXG_TILE", "FXG_BLOCKS_PER_CU", "FXG_TICKET_GROUPS"):
        os.environ.pop(k, None)
    os.environ.update(env)
    ms = []
    for _ in range(reps):
        eng.run(b, q, P, fixed_len=150, compact=compact, meta=False, outputs=outs if compact else None)
        ms.append(eng.last_kernel_ms())
    li = eng.last_launch()
    print(json.dumps(dict(label=label, ms_min=round(min(ms), 3), ms_avg=round(sum(ms) / len(ms), 3), grid=li["grid"], tile=li["tile_reads"], lds=li["lds"])), flush=True)


cfgs = json.loads(os.environ.get("ABLATE", "null")) or [
    ["full", {}],
    ["decision-only", {}, False],
    ["no-gather", {"FXG_DEBUG": "1"}],
    ["no-lookback", {"FXG_DEBUG": "2"}],
    ["no-qual-gather", {"FXG_DEBUG": "4"}],
    ["no-bitmaps", {"FXG_DEBUG": "8"}],
    ["no-gather,no-lookback", {"FXG_DEBUG": "3"}],
    ["no-gather,no-lookback,no-bitmaps", {"FXG_DEBUG": "11"}],
    ["tile128", {"FXG_TILE": "128"}],
    ["tile64", {"FXG_TILE": "64"}],
    ["bpc4", {"FXG_BLOCKS_PER_CU": "4"}],
    ["bpc2", {"FXG_BLOCKS_PER_CU": "2"}],
    ["bpc1", {"FXG_BLOCKS_PER_CU": "1"}],
    ["tile128 bpc4", {"FXG_TILE": "128", "FXG_BLOCKS_PER_CU": "4"}],
]
for c in cfgs:
    t(c[0], c[1], *(c[2:] or [True]))
