#!/bin/bash
# round 6, call at (as again, with the odd-length kernel): the statistics kernel's piece form (dense even-length rows: aligned 16-byte pieces of the byte stream, no padding bytes) against the row-strip form
# of the same library (FXG_QS_ROUND_ROBIN=3): parity, time at 150 and other lengths, FETCH_SIZE / WRITE_SIZE
O=gpurun_out/r06at; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "quality_stats_vs_oracle or long_reads" 2>&1 | tail -n 3 | tee $O/stats_piece_parity.txt
for rep in 1 2; do for k in 1 3; do
  echo -n "order=$k L=150: "; FXG_QS_ROUND_ROBIN=$k timeout 300 python scripts/bench_stats.py 2>&1 | tail -n 1 | cut -c1-110
done; done | tee $O/stats_piece_vs_rows.txt
for L in 36 51 75 76 100 101 126 151 160; do for k in 1 3; do
  echo -n "order=$k L=$L: "; LEN=$L FXG_QS_ROUND_ROBIN=$k timeout 300 python scripts/bench_stats.py 2>&1 | tail -n 1 | cut -c1-110
done; done | tee $O/stats_piece_vs_rows_by_length.txt
