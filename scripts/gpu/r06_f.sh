#!/bin/bash
# round 6, call f: the gap penalty as an SGPR / VGPR operand, the scheduling fence, the phase clocks of the new kernel
O=gpurun_out/r06f; mkdir -p $O
LIBS=fastx_toolkit_amd/libfxg.so,fastx_toolkit_amd/libfxg_v_m5s.so,fastx_toolkit_amd/libfxg_v_m5v.so,fastx_toolkit_amd/libfxg_v_sched.so timeout 900 python scripts/clip_ab.py > $O/clip_ab.txt 2>&1
cut -c1-200 $O/clip_ab.txt
for c in cfg3 cfg5; do CFG=$c LIBS=fastx_toolkit_amd/libfxg_v_abl.so timeout 300 python scripts/ablate_clip.py; done > $O/phase_clocks.txt 2>&1
cat $O/phase_clocks.txt
for c in cfg3 cfg5; do for d in 32 16 48; do FXG_DEBUG=$d CFG=$c LIBS=fastx_toolkit_amd/libfxg_v_abl.so timeout 300 python scripts/ablate_clip.py | head -1 | cut -c1-120 | sed "s/^/FXG_DEBUG=$d /"; done; done > $O/dp_alone.txt 2>&1
cat $O/dp_alone.txt
