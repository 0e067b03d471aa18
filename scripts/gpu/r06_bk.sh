#!/bin/bash
# round 6, call bk: the reverse complement's dword-aligned window loads switched on per launch by the plan (HEAD) against forced off / on (FXG_REV_DW=0 / 1):
# parity of the reverse-complement paths in all three, the row-length table, cfg4
O=gpurun_out/r06bk; mkdir -p $O
for k in "" 1; do
  FXG_REV_DW=$k timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli.py -q -m gpu -k "fuzz or bad_base or scan_timeout or galaxy or long_reads or cfg4 or configs or full" 2>&1 | tail -n 1 | sed "s/^/FXG_REV_DW=$k /"
done | tee $O/rev_parity.txt
for k in auto 0 1; do
  [ $k = auto ] && unset FXG_REV_DW || export FXG_REV_DW=$k
  CASES=2 timeout 900 python scripts/gather_alignment.py 2>&1 | grep "^{" | grep "reverse" | sed "s/^/dw=$k /" | cut -c1-250
  CASES=3 timeout 900 python scripts/gather_alignment.py 2>&1 | grep "^{" | sed "s/^/dw=$k /" | cut -c1-250
done | tee $O/rev_dword_loads_by_plan.txt
unset FXG_REV_DW
