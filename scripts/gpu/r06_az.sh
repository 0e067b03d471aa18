#!/bin/bash
# round 6, call az: the census with four dword loads in flight (HEAD) against one at a time (libfxg_v_census1.so) and one byte per step (libfxg_v_nogrid.so)
O=gpurun_out/r06az; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "fuzz or bad_base or scan_timeout or long_reads" 2>&1 | tail -n 2 | tee $O/census_parity.txt
for v in libfxg.so libfxg_v_census1.so libfxg_v_nogrid.so libfxg.so libfxg_v_census1.so libfxg_v_nogrid.so; do
  FXG_LIB=$PWD/fastx_toolkit_amd/$v timeout 900 python scripts/bench_stages.py 2>&1 | grep "^{" | grep "artifacts\|fasta" | sed "s/^/$v /" | cut -c1-250
done | tee $O/census_swar_vs_bytes.txt
