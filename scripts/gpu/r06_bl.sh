#!/bin/bash
# round 6, call bl: the streaming tile kernels (fixed trimmer, reverse complement, masker, census) by tile size (FXG_TILE = 64 / 128 (the plan's choice for 150-byte rows) / 256)
O=gpurun_out/r06bl; mkdir -p $O
for t in 128 256 64 128 256; do
  FXG_TILE=$t timeout 900 python scripts/bench_stages.py 2>&1 | grep "^{" | grep -v "quality_" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('tile $t', d['tool'].ljust(28), d['ms_min'], d['frac_hbm'])"
done | tee $O/stages_by_tile.txt
