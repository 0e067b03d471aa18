#!/bin/bash
# round 6, call bn: the reverse complement's instance with dword-aligned window loads (fxg_kernel_tiles<0,5>), picked by the plan where no source window of the launch
# is dword aligned: parity of the reverse-complement paths with the plan's choice and with the instance forced on every launch (FXG_REV_DW=1: ragged batches, every
# trim range), then the row-length table with the plan's choice against the instance forced off (FXG_REV_DW=0)
O=gpurun_out/r06bn; mkdir -p $O
for k in "" 1; do
  FXG_REV_DW=$k timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli.py -q -m gpu -k "fuzz or bad_base or scan_timeout or galaxy or long_reads or cfg4 or configs or full" 2>&1 | tail -n 1 | sed "s/^/FXG_REV_DW=$k /"
done | tee $O/rev_parity.txt
for k in auto 0 auto 0; do
  [ $k = auto ] && unset FXG_REV_DW || export FXG_REV_DW=$k
  CASES=2 timeout 900 python scripts/gather_alignment.py 2>&1 | grep "^{" | grep "reverse" | sed "s/^/dw=$k /" | cut -c1-250
  CASES=3 timeout 900 python scripts/gather_alignment.py 2>&1 | grep "^{" | sed "s/^/dw=$k /" | cut -c1-250
done | tee $O/rev_dw_instance.txt
unset FXG_REV_DW
