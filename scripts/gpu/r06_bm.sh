#!/bin/bash
# round 6, call bm: cfg2 once more in one call: HEAD against the write-out with default-policy stores (-DFXG_V_NO_NTS: re-measured now that the units are dealt on the
# line grid) and against the write-out without the line grid (-DFXG_STORE_GRID=0)
O=gpurun_out/r06bm; mkdir -p $O
P=$PWD/fastx_toolkit_amd
one() { FXG_LIB=$P/$2 timeout 600 python bench.py --config $1 --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --headline-only 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1 $2 ms_per_step', d['ms_per_step'], 'kernel_ms', r.get('kernel_ms_avg'), 'frac', r['frac'], 'self_check', d.get('self_check',{}).get('matches_pinned'))"; }
for rep in 1 2 3 4; do for v in libfxg.so libfxg_v_nonts.so libfxg_v_nogrid.so; do one cfg2 $v; done; done | tee $O/cfg2_stores.txt
