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
eng.sync()
KNOBS = ("FXG_DEBUG", "FXG_ROWS", "FXG_TILE", "FXG_BLOCKS_PER_CU", "FXG_TICKET_GROUPS", "FXG_QLDS", "FXG_QLDS_BUDGET")


def t(label, env, compact=True, reps=4):
    for k in KNOBS:
        os.environ.pop(k, None)
    os.environ.update(env)
    e = Engine(0)                                  # the context reads its tuning knobs when it is created
    e.set_profiling(True)
    ms = []
    for _ in range(reps):
        r = e.run(b, q, P, fixed_len=150, compact=compact, meta=False, outputs=outs if compact else None)
        ms.append(e.last_kernel_ms())
    c = r.counters
    li = e.last_launch()
    if hasattr(e.lib, "fxg_debug_phase_clocks") and compact:
        import ctypes
        ph = (ctypes.c_uint64 * 11)()
        e.lib.fxg_debug_phase_clocks(e.ctx, ph)
        waves = li["grid"] - 1
        names = ("fetch-q", "decide", "wait", "flush-q", "fetch-b", "flush-b", "pack-q", "read+pack-b")
        if "rows" in li["kernel"]:
            print("   per wave, ms: " + "  ".join("%s %.2f" % (nm, x / 1e5 / waves) for nm, x in zip(names, ph)), flush=True)
        if ph[8]:
            print("   scanner: %d rounds, %.0f tiles/round, load wait %.2f us/round, scan %.2f us/round, total %.2f ms" %
                  (ph[8], (R + 63) // 64 / ph[8], ph[9] / 100.0 / ph[8], ph[10] / 100.0 / ph[8], (ph[9] + ph[10]) / 1e5), flush=True)
    print(json.dumps(dict(label=label, ms_min=round(min(ms), 3), ms_avg=round(sum(ms) / len(ms), 3), grid=li["grid"], tile=li["tile_reads"], lds=li["lds"],
                          kept=int(c[1]), kept_bases=int(c[2]), err=int(c[15]))), flush=True)
    e.close()


cfgs = json.loads(os.environ.get("ABLATE", "null")) or [
    ["full", {}],
    ["decision-only", {}, False],
    ["qlds0 t256", {"FXG_QLDS": "0"}],
    ["qlds1 52K", {"FXG_QLDS": "1", "FXG_QLDS_BUDGET": "53248"}],
    ["qlds1 26K", {"FXG_QLDS": "1", "FXG_QLDS_BUDGET": "26624"}],
]
for c in cfgs:
    t(c[0], c[1], *(c[2:] or [True]))
