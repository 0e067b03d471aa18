#!/usr/bin/env python3
"""GPU box: what the clip kernel's DP alone reaches with 3 / 4 workgroups per CU (decision-only launches) next to the full launch --
the headroom a design whose DP workgroups never wait for the write-out can have."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT)
    from fastx_toolkit_amd import Engine, make_params
    cfg, compact, R = sys.argv[1], sys.argv[2] == "1", int(os.environ.get("READS", "20000000"))
    AD = b"AGATCGGAAGAGC"
    eng = Engine(0)
    L = 100 if cfg == "cfg3" else 150
    b, q = eng.synth(3 if cfg == "cfg3" else 5, 0, R, L, True)
    P = (make_params(stages=1, adapter=AD, clip_min_len=15, clip_flags=4) if cfg == "cfg3" else
         make_params(stages=7, adapter=AD, clip_min_len=15, clip_flags=4, qt_threshold=20, qt_min_len=30, qf_min_quality=20, qf_min_percent=80))
    outs = eng.alloc_outputs(R, L, compact=compact, meta=False)
    eng.set_profiling(True)
    ms = []
    for _ in range(4):
        eng.run(b, q, P, fixed_len=L, compact=compact, meta=False, outputs=outs)
        ms.append(eng.last_kernel_ms())
    li = eng.last_launch()
    print(json.dumps(dict(cfg=cfg, compact=compact, blocks_per_cu=os.environ.get("FXG_BLOCKS_PER_CU"), roles=os.environ.get("FXG_CLIP_WRITER_EVERY"), reads=R, ms_min=round(min(ms), 3), grid=li["grid"], lds=li["lds"], kernel=li["kernel"].split()[0])))
else:
    for cfg in ("cfg3", "cfg5"):
        for compact, bpc in (("1", None), ("0", "4"), ("0", "3"), ("0", "2")):
            env = dict(os.environ)
            if bpc:
                env["FXG_BLOCKS_PER_CU"] = bpc
            subprocess.run([sys.executable, os.path.abspath(__file__), cfg, compact], env=env)
