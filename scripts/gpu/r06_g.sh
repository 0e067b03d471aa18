#!/bin/bash
# round 6, call g: batched staging loads (A/B), and the product's own timing with one phase taken out at a time (-DFXG_DBG_BITS: no phase clocks)
O=gpurun_out/r06g; mkdir -p $O
LIBS=fastx_toolkit_amd/libfxg_v_pre.so,fastx_toolkit_amd/libfxg.so timeout 900 python scripts/clip_ab.py > $O/clip_ab.txt 2>&1
cut -c1-200 $O/clip_ab.txt
LIBS=fastx_toolkit_amd/libfxg_v_dbg.so DEBUGS=0,1,2,3,16,19,32,48,51 REPS=3 timeout 1200 python scripts/clip_ab.py > $O/phases_out.txt 2>&1
LIBS=fastx_toolkit_amd/libfxg_v_dbg.so DEBUGS=0,16,32,48 COMPACT=0 REPS=3 timeout 1200 python scripts/clip_ab.py >> $O/phases_out.txt 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r06g/phases_out.txt"):
    if l.startswith("{"):
        d = json.loads(l); print(d["cfg"], "debug", d["debug"], "compact", d["compact"], "ms", d["ms_min"])
    else: print(l.strip()[:200])
PY
