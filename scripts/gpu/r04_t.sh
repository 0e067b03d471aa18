#!/bin/bash
# round 4, call t: phase clocks of the clip kernel as it is now (ticket after the decision, depth 4, uniform loops)
mkdir -p gpurun_out/r04t
for c in cfg3 cfg5; do LIBS=fastx_toolkit_amd/libfxg_x_abl.so CFG=$c timeout 280 python scripts/ablate_clip.py 2>&1 | grep -v "amdgpu.ids"; done | tee gpurun_out/r04t/ablate_clip.txt
for c in cfg3 cfg5; do FXG_CLIP_DEPTH_RT=2 LIBS=fastx_toolkit_amd/libfxg_x_abl.so CFG=$c timeout 280 python scripts/ablate_clip.py 2>&1 | grep -v "amdgpu.ids\|scanner" | sed 's/^/depth2 /'; done | tee -a gpurun_out/r04t/ablate_clip.txt
