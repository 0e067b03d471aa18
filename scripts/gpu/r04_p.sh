#!/bin/bash
# round 4, call p: register budget of the clip instances now that every budget is correct: time per adapter length at the shipped waves,
# with the wide buckets at three waves, and with every instance at three / four waves
mkdir -p gpurun_out/r04p
for v in ship wide3 all3 mid4; do
  lib=fastx_toolkit_amd/libfxg_x_$v.so; [ $v = ship ] && lib=fastx_toolkit_amd/libfxg.so
  for L in 100 150; do
    FXG_LIB=$lib L=$L READS=10000000 python scripts/clip_by_adapter_len.py 13 16 20 24 28 32 34 40 48 49 64 65 99 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$v L=$L A=%d %s %.3f ms %.0f GCUPS' % (d['adapter_len'], d['kernel'].split()[0], d['ms_min'], d['gcups']))"
  done
done | tee gpurun_out/r04p/waves_by_adapter_len.txt
