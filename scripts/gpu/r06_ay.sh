#!/bin/bash
# round 6, call ay: the base census four bases per step (artifacts filter, fastq_to_fasta's N filter): parity (fuzz, bad bases, scan recovery, CLI vs the real
# libfastx), then the stage table with HEAD and with the previous census (libfxg_v_nogrid.so: the library of call au, one byte per step)
O=gpurun_out/r06ay; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli.py -q -m gpu -k "fuzz or bad_base or scan_timeout or galaxy or long_reads or text" 2>&1 | tail -n 3 | tee $O/census_parity.txt
for v in libfxg.so libfxg_v_nogrid.so libfxg.so libfxg_v_nogrid.so; do
  FXG_LIB=$PWD/fastx_toolkit_amd/$v timeout 900 python scripts/bench_stages.py 2>&1 | grep "^{" | grep "artifacts\|fasta" | sed "s/^/$v /" | cut -c1-250
done | tee $O/census_swar_vs_bytes.txt
timeout 900 python scripts/bench_stages.py 2>&1 | grep "^{" > $O/stages_50M_x150.txt
