#!/bin/bash
# round 4, call w: the DP alone -- decision-only launches that stage a workgroup's first tile once and decide it again for every ticket
mkdir -p gpurun_out/r04w
for c in cfg3 cfg5; do for bpc in 4 3 2 1; do for dbg in 0 24; do
  [ $c = cfg5 ] && [ $bpc = 4 ] && continue
  echo "blocks_per_cu $bpc debug $dbg $(COMPACT=0 FXG_BLOCKS_PER_CU=$bpc FXG_DEBUG=$dbg LIBS=fastx_toolkit_amd/libfxg_x_abl.so CFG=$c timeout 280 python scripts/ablate_clip.py 2>&1 | grep -v "amdgpu.ids\|scanner\|barrier" | tr '\n' ' ' | cut -c1-330)"
done; done; done | tee gpurun_out/r04w/dp_alone.txt
