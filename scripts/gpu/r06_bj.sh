#!/bin/bash
# round 6, call bj: the reverse complement's dword-aligned window loads with the fifth word only where the window is off a dword boundary (HEAD) against always
# (libfxg_v_revdw5.so) and plain 16-byte loads (libfxg_v_revld.so)
O=gpurun_out/r06bj; mkdir -p $O
P=$PWD/fastx_toolkit_amd
for v in libfxg.so libfxg_v_revdw5.so libfxg_v_revld.so; do
  CASES=2 FXG_LIB=$P/$v timeout 900 python scripts/gather_alignment.py 2>&1 | grep "^{" | grep "reverse" | sed "s/^/$v /" | cut -c1-250
  CASES=3 FXG_LIB=$P/$v timeout 900 python scripts/gather_alignment.py 2>&1 | grep "^{" | sed "s/^/$v /" | cut -c1-250
done | tee $O/rev_dword_loads.txt
one() { FXG_LIB=$P/$2 timeout 600 python bench.py --config $1 --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --headline-only 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1 $2 ms_per_step', d['ms_per_step'], 'kernel_ms', r.get('kernel_ms_avg'), 'frac', r['frac'], 'self_check', d.get('self_check',{}).get('matches_pinned'))"; }
for rep in 1 2; do for v in libfxg.so libfxg_v_revdw5.so libfxg_v_revld.so; do one cfg4 $v; done; done | tee $O/rev_dword_loads_cfg4.txt
