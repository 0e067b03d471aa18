#!/bin/bash
# round 4, call u: start the workgroups of a CU a fraction of a tile apart (they run identical phases in lock-step otherwise)
mkdir -p gpurun_out/r04u
for us in 0 5 10 15 20 25 30 40; do
  for cfg in cfg3 cfg5; do echo "stagger_us $us $(FXG_STAGGER_US=$us FXG_LIB=fastx_toolkit_amd/libfxg_x_stagger.so python scripts/clip_roles_potential.py $cfg 1 2>&1 | tail -1 | cut -c1-140)"; done
done | tee gpurun_out/r04u/stagger.txt
for us in 0 15 25; do for c in cfg3; do FXG_STAGGER_US=$us LIBS=fastx_toolkit_amd/libfxg_x_abl.so CFG=$c timeout 280 python scripts/ablate_clip.py 2>&1 | grep -v "amdgpu.ids\|scanner" | sed "s/^/stagger $us /"; done; done | tee gpurun_out/r04u/ablate_stagger.txt
