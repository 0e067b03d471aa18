#!/bin/bash
# round 6, call ac: clip kernel with s_setprio around the write-out (3, 1) and around the staging (2): cfg3 / cfg5, 20 M reads, two rounds
O=gpurun_out/r06ac; mkdir -p $O
L=fastx_toolkit_amd/libfxg.so,fastx_toolkit_amd/libfxg_v_prw3.so,fastx_toolkit_amd/libfxg_v_prw1.so,fastx_toolkit_amd/libfxg_v_prws3.so
LIBS=$L,$L CFGS=cfg3,cfg5 READS=20000000 REPS=7 timeout 900 python scripts/clip_ab.py > $O/clip_ab_setprio.txt 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r06ac/clip_ab_setprio.txt"):
    if l.startswith("{"):
        d = json.loads(l); print(d["lib"].split("/")[-1], d["cfg"], d["ms_min"], d["ms_med"], d["checksum"])
    else: print(l[:200].rstrip())
PY
