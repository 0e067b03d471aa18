#!/bin/bash
# round 6, call a: pass 1 of the clip DP with its pair values out of an LDS table (v_fma_mix_f32) against the compare + select cell of round 5
O=gpurun_out/r06a; mkdir -p $O
LIBS=fastx_toolkit_amd/libfxg_v_noptab.so,fastx_toolkit_amd/libfxg.so,fastx_toolkit_amd/libfxg_v_sched.so timeout 900 python scripts/clip_ab.py > $O/clip_ab.txt 2>&1
cat $O/clip_ab.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "clip or fuzz or config" > $O/pytest_clip.txt 2>&1; tail -n 5 $O/pytest_clip.txt
