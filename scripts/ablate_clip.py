#!/usr/bin/env python3
"""Phase clocks of the clip kernel (fxg_kernel_tiles<-13,0>) on the GPU box (not a test): -DFXG_ABLATION builds, libraries named in LIBS."""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
libs = os.environ.get("LIBS", "build/libfxg_abl64.so,build/libfxg_abl256.so").split(",")
CFG = os.environ.get("CFG", "cfg3")
R = int(os.environ.get("READS", "20000000"))
AD = b"AGATCGGAAGAGC"
for lib in libs:
    os.environ["FXG_LIB"] = os.path.join(ROOT, lib)
    import importlib
    import fastx_toolkit_amd.engine as E
    E._LIB = None
    importlib.reload(E)
    eng = E.Engine(0)
    L = 100 if CFG == "cfg3" else 150
    b, q = eng.synth(3 if CFG == "cfg3" else 5, 0, R, L, True)
    P = (E.make_params(stages=1, adapter=AD, clip_min_len=15, clip_flags=4) if CFG == "cfg3" else
         E.make_params(stages=7, adapter=AD, clip_min_len=15, clip_flags=4, qt_threshold=20, qt_min_len=30, qf_min_quality=20, qf_min_percent=80))
    COMPACT = os.environ.get("COMPACT", "1") != "0"
    outs = eng.alloc_outputs(R, L, compact=COMPACT, meta=False)
    eng.set_profiling(True)
    ms = []
    for _ in range(3):
        r = eng.run(b, q, P, fixed_len=L, compact=COMPACT, meta=False, outputs=outs)
        ms.append(eng.last_kernel_ms())
    li = eng.last_launch()
    ph = (ctypes.c_uint64 * 11)()
    eng.lib.fxg_debug_phase_clocks(eng.ctx, ph)
    wgs = li["grid"] - (1 if COMPACT else 0)
    names = ("stage", "decide(DP)", "scan+publish", "wait prefix", "meta+gather")
    print(json.dumps(dict(lib=lib, cfg=CFG, compact=COMPACT, reads=R, ms_min=round(min(ms), 3), grid=li["grid"], block=li["block"], tile=li["tile_reads"], lds=li["lds"])))
    print("   per workgroup, ms: " + "  ".join("%s %.3f" % (nm, x / 1e5 / wgs) for nm, x in zip(names, ph)), flush=True)
    if ph[7]:
        print("   shader clock over the kernel: %.0f MHz (s_memtime %d ticks in %d x 10 ns)" % (ph[6] / ph[7] * 100.0, ph[6], ph[7]), flush=True)
    if ph[5]:
        print("   barrier after the decision: %.3f ms per WAVE (of the decide time above; 4 waves per workgroup)" % (ph[5] / 1e5 / wgs / (li["block"] // 64)), flush=True)
    if ph[8]:
        print("   scanner: %d rounds, %.1f tiles/round, load wait %.2f us/round, scan %.2f us/round, total %.2f ms" %
              (ph[8], (R + li["tile_reads"] - 1) // li["tile_reads"] / ph[8], ph[9] / 100.0 / ph[8], ph[10] / 100.0 / ph[8], (ph[9] + ph[10]) / 1e5), flush=True)
    eng.close()
    del eng
