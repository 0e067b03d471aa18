#!/usr/bin/env python3
"""A/B of engine builds on the clip configs (GPU box, not a test): kernel ms of cfg3 / cfg5 per library + a checksum of everything the launch wrote.

    LIBS=fastx_toolkit_amd/libfxg.so,fastx_toolkit_amd/libfxg_v_noptab.so CFGS=cfg3,cfg5 READS=20000000 python scripts/clip_ab.py
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
AD = os.environ.get("ADAPTER", "AGATCGGAAGAGC").encode()

if len(sys.argv) > 1 and sys.argv[1] == "one":
    from fastx_toolkit_amd import Engine, make_params
    cfg, R = sys.argv[2], int(sys.argv[3])
    eng = Engine(0)
    L = int(os.environ.get("READ_LEN", "100" if cfg == "cfg3" else "150"))
    b, q = eng.synth(3 if cfg == "cfg3" else 5, 0, R, L, True)
    P = (make_params(stages=1, adapter=AD, clip_min_len=15, clip_flags=4) if cfg == "cfg3" else
         make_params(stages=7, adapter=AD, clip_min_len=15, clip_flags=4, qt_threshold=20, qt_min_len=30, qf_min_quality=20, qf_min_percent=80))
    COMPACT = os.environ.get("COMPACT", "1") != "0"
    outs = eng.alloc_outputs(R, L, compact=COMPACT, meta=False)
    eng.set_profiling(True)
    ms = []
    for _ in range(int(os.environ.get("REPS", "5"))):
        r = eng.run(b, q, P, fixed_len=L, compact=COMPACT, meta=False, outputs=outs)
        ms.append(eng.last_kernel_ms())
    li = eng.last_launch()
    print(json.dumps(dict(lib=os.environ.get("FXG_LIB", "default"), cfg=cfg, reads=R, L=L, alen=len(AD), ms_min=round(min(ms), 3), ms_med=round(sorted(ms)[len(ms) // 2], 3),
                          kernel=li.get("kernel"), lds=li["lds"], tile=li["tile_reads"], debug=os.environ.get("FXG_DEBUG", ""), compact=COMPACT,
                          kept=int(r.kept), kept_bases=int(r.kept_bytes), checksum=int(r.checksum()) if COMPACT else None)), flush=True)
    eng.close()
else:
    libs = os.environ.get("LIBS", "fastx_toolkit_amd/libfxg.so").split(",")
    for cfg in os.environ.get("CFGS", "cfg3,cfg5").split(","):
        for lib in libs:
          for dbg in os.environ.get("DEBUGS", "").split(",") or [""]:      # FXG_DEBUG values of a -DFXG_DBG_BITS build (timing with a phase taken out; wrong results)
            env = dict(os.environ, FXG_LIB=os.path.join(ROOT, lib))
            if dbg:
                env["FXG_DEBUG"] = dbg
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "one", cfg, os.environ.get("READS", "20000000")], env=env, capture_output=True, text=True, timeout=600)
            print(p.stdout.strip() or ("FAILED %s %s: %s" % (lib, cfg, p.stderr[-600:])), flush=True)
