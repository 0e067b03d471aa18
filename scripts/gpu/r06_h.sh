#!/bin/bash
# round 6, call h: what pass 1 of the clip DP is waiting for (decision-only launches, pass 1 alone, staging of the first tile only: FXG_DEBUG=48)
O=gpurun_out/r06h; mkdir -p $O
LIBS=fastx_toolkit_amd/libfxg_v_d_base.so,fastx_toolkit_amd/libfxg_v_d_sched.so,fastx_toolkit_amd/libfxg_v_d_nofetch.so,fastx_toolkit_amd/libfxg_v_d_nolut.so DEBUGS=48 COMPACT=0 REPS=3 timeout 1200 python scripts/clip_ab.py > $O/pass1_alone.txt 2>&1
LIBS=fastx_toolkit_amd/libfxg_v_d_base.so,fastx_toolkit_amd/libfxg_v_d_sched.so DEBUGS=0 REPS=3 timeout 1200 python scripts/clip_ab.py >> $O/pass1_alone.txt 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r06h/pass1_alone.txt"):
    if l.startswith("{"):
        d = json.loads(l); print(d["lib"].split("libfxg_v_")[-1], d["cfg"], "debug", d["debug"], "compact", d["compact"], "ms", d["ms_min"], d["ms_med"])
    else: print(l.strip()[:300])
PY
