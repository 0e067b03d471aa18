#!/bin/bash
# round 6, call v: A/B of the write-out's rank search with a starting guess (libfxg.so) against the plain bisection (libfxg_v_norg.so = -DFXG_NO_RANK_GUESS)
O=gpurun_out/r06v; mkdir -p $O
LIBS=fastx_toolkit_amd/libfxg.so,fastx_toolkit_amd/libfxg_v_norg.so,fastx_toolkit_amd/libfxg.so,fastx_toolkit_amd/libfxg_v_norg.so CFGS=cfg3,cfg5 READS=20000000 REPS=7 timeout 600 python scripts/clip_ab.py > $O/clip_ab_rank_guess.txt 2>&1
cut -c1-400 $O/clip_ab_rank_guess.txt
