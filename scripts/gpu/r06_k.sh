#!/bin/bash
# round 6, call k: the statistics kernel's loads alone / without bank conflicts; the clip kernel at HEAD against the round-5 cell; the GPU tier
O=gpurun_out/r06k; mkdir -p $O
for v in qs1 default qsnoacc1 qsnoacc qsfake; do lib=fastx_toolkit_amd/libfxg_v_$v.so; [ $v = default ] && lib=fastx_toolkit_amd/libfxg.so
  echo -n "$v: "; FXG_LIB=$PWD/$lib timeout 300 python scripts/bench_stats.py 2>/dev/null | tail -1 | cut -c1-120; done > $O/stats_experiments.txt 2>&1
cat $O/stats_experiments.txt
LIBS=fastx_toolkit_amd/libfxg_v_noptab.so,fastx_toolkit_amd/libfxg.so timeout 900 python scripts/clip_ab.py > $O/clip_ab.txt 2>&1
cut -c1-200 $O/clip_ab.txt
timeout 2400 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_clip_matrix.py --durations=8 > $O/pytest_gpu.txt 2>&1; tail -n 16 $O/pytest_gpu.txt
