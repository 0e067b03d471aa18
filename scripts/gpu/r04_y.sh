#!/bin/bash
# round 4, call y: pass 1 alone against both passes (decision-only launches on a resident tile, 4 and 1 workgroups per CU)
mkdir -p gpurun_out/r04y
for c in cfg3; do for bpc in 4 2 1; do for dbg in 24 56; do
  echo "blocks_per_cu $bpc debug $dbg $(COMPACT=0 FXG_BLOCKS_PER_CU=$bpc FXG_DEBUG=$dbg LIBS=fastx_toolkit_amd/libfxg_x_abl.so CFG=$c timeout 280 python scripts/ablate_clip.py 2>&1 | grep -v "amdgpu.ids\|scanner\|barrier\|shader clock" | tr '\n' ' ' | cut -c100-330)"
done; done; done | tee gpurun_out/r04y/pass1_alone.txt
