#!/bin/bash
# round 4, call q: the 100-column bucket at two and three waves (65- and 99-base adapters, with and without N); clip parity at the new budgets
mkdir -p gpurun_out/r04q
for v in ship wide3; do
  lib=fastx_toolkit_amd/libfxg_x_$v.so; [ $v = ship ] && lib=fastx_toolkit_amd/libfxg.so
  for wn in 0 1; do for L in 100 150; do
    if [ $wn = 1 ]; then export WITH_N=1; else unset WITH_N; fi
    FXG_LIB=$lib L=$L READS=10000000 python scripts/clip_by_adapter_len.py 36 49 64 65 99 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$v N=$wn L=$L A=%d %s %.3f ms %.0f GCUPS' % (d['adapter_len'], d['kernel'].split()[0], d['ms_min'], d['gcups']))"
  done; done
done | tee gpurun_out/r04q/wide_bucket_waves.txt
unset WITH_N
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "fuzz or adversarial or history or configs_vs" 2>&1 | tail -3
