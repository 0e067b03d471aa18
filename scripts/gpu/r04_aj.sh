#!/bin/bash
# round 4, call aj: the over-the-batch clip form as kernels of its own: A/B against the library before it, whole GPU tier, closing evidence
mkdir -p gpurun_out/r04aj
for i in 1 2; do for v in pre_gl head; do lib=fastx_toolkit_amd/libfxg_x_$v.so; [ $v = head ] && lib=fastx_toolkit_amd/libfxg.so; for cfg in cfg3 cfg5; do echo "$v $(FXG_LIB=$lib python scripts/clip_roles_potential.py $cfg 1 2>&1 | tail -1 | cut -c1-100)"; done; done; done | tee gpurun_out/r04aj/ab_pre_gl_vs_head.txt
SHAPES=300:6000000:1,1000:2000000:1 python scripts/debug/clip_global_vs_staged.py 2>&1 | grep -v amdgpu | cut -c1-400 | tee gpurun_out/r04aj/clip_global.txt
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r04aj/pytest_gpu.txt 2>&1; grep -a "passed\|failed" gpurun_out/r04aj/pytest_gpu.txt | tail -2
bash scripts/gpu/r04_final.sh > gpurun_out/r04aj/final.log 2>&1; grep "traffic_over\|rc=" gpurun_out/r04aj/final.log | tail -12
