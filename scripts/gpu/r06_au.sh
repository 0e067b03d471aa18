#!/bin/bash
# round 6, call au: the write-outs' 16-byte units dealt to lanes by their place in the 128-byte line (HEAD, FXG_STORE_GRID=1) against by their place in the tile's
# output (libfxg_v_nogrid.so): the bench line of each config, alternating; every line self-checks its pinned checksum
O=gpurun_out/r06au; mkdir -p $O
P=$PWD/fastx_toolkit_amd
one() { FXG_LIB=$P/$2 timeout 600 python bench.py --config $1 --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --headline-only 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1 $2 ms_per_step', d['ms_per_step'], 'kernel_ms', r.get('kernel_ms_avg'), 'frac', r['frac'], 'self_check', d.get('self_check',{}).get('matches_pinned'))"; }
for rep in 1 2 3 4; do for v in libfxg.so libfxg_v_nogrid.so; do one cfg2 $v; done; done | tee $O/store_grid_cfg2.txt
for c in cfg4 cfg3 cfg5shard; do for rep in 1 2; do for v in libfxg.so libfxg_v_nogrid.so; do one $c $v; done; done; done | tee $O/store_grid_other.txt
