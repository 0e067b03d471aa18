#!/bin/bash
# round 6, call bi: the reverse complement's source windows through dword-aligned loads + funnel shift (HEAD) against plain 16-byte loads (libfxg_v_revld.so):
# parity of the reverse-complement paths, the row-length table, cfg4's bench line
O=gpurun_out/r06bi; mkdir -p $O
P=$PWD/fastx_toolkit_amd
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli.py -q -m gpu -k "fuzz or bad_base or scan_timeout or galaxy or long_reads or cfg4 or configs or full" 2>&1 | tail -n 3 | tee $O/rev_parity.txt
for v in libfxg.so libfxg_v_revld.so; do
  CASES=2 FXG_LIB=$P/$v timeout 900 python scripts/gather_alignment.py 2>&1 | grep "^{" | grep "reverse\|revcomp" | sed "s/^/$v /" | cut -c1-250
  CASES=3 FXG_LIB=$P/$v timeout 900 python scripts/gather_alignment.py 2>&1 | grep "^{" | sed "s/^/$v /" | cut -c1-250
done | tee $O/rev_dword_loads.txt
one() { FXG_LIB=$P/$2 timeout 600 python bench.py --config $1 --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --headline-only 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1 $2 ms_per_step', d['ms_per_step'], 'kernel_ms', r.get('kernel_ms_avg'), 'frac', r['frac'], 'self_check', d.get('self_check',{}).get('matches_pinned'))"; }
for rep in 1 2 3; do for v in libfxg.so libfxg_v_revld.so; do one cfg4 $v; done; done | tee $O/rev_dword_loads_cfg4.txt
