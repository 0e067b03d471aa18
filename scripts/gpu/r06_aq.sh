#!/bin/bash
# round 6, call aq: the statistics kernel without the tail masks on dense rows (-DFXG_QS_NOMASK), with the bank swizzle by strip alone (-DFXG_QS_SWZ_STRIP_ONLY), both;
# the shipped kernel on rows of 144 / 160 bytes (every 16-byte piece aligned, no tail) against 150 -- what alignment and the tail are worth
O=gpurun_out/r06aq; mkdir -p $O
P=$PWD/fastx_toolkit_amd
for v in libfxg_v_qsnm.so libfxg_v_qsnmsw.so; do
  echo -n "$v parity: "; FXG_LIB=$P/$v timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "quality_stats_vs_oracle" 2>&1 | tail -n 1
done | tee $O/stats_variants_parity.txt
for rep in 1 2 3 4; do for v in libfxg.so libfxg_v_qsnm.so libfxg_v_qssw.so libfxg_v_qsnmsw.so; do
  echo -n "$v: "; FXG_LIB=$P/$v timeout 300 python scripts/bench_stats.py 2>&1 | tail -n 1 | cut -c1-110
done; done | tee $O/stats_nomask_swz.txt
for rep in 1 2 3; do for L in 150 144 160 128; do for v in libfxg.so libfxg_v_qsnm.so; do
  echo -n "$v L=$L: "; LEN=$L FXG_LIB=$P/$v timeout 300 python scripts/bench_stats.py 2>&1 | tail -n 1 | cut -c1-110
done; done; done | tee $O/stats_by_row_length.txt
