#!/bin/bash
# round 6, call bb (ba again; HEAD = the bitmaps in the layout's bitmap region, tiles of 20 KB, granule table; libfxg_v_census2.so = the bitmaps in the staged tile's region): the census from four bitmaps of the tile (phase 1 streams the rows; no staged tile, no walk) against four bases per step from the staged tile
# (libfxg_v_census1.so) and one byte per step (libfxg_v_nogrid.so): parity, then the two tools' kernel time
O=gpurun_out/r06bb; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli.py -q -m gpu -k "fuzz or bad_base or scan_timeout or galaxy or long_reads or text" 2>&1 | tail -n 3 | tee $O/census_parity.txt
for v in libfxg.so libfxg_v_census2.so libfxg_v_census1.so libfxg_v_nogrid.so libfxg.so libfxg_v_census2.so libfxg_v_census1.so libfxg_v_nogrid.so; do
  FXG_LIB=$PWD/fastx_toolkit_amd/$v timeout 900 python scripts/bench_stages.py 2>&1 | grep "^{" | grep "artifacts\|fasta" | sed "s/^/$v /" | cut -c1-250
done | tee $O/census_bitmaps_vs_walk.txt
timeout 900 python scripts/bench_stages.py 2>&1 | grep "^{" > $O/stages_50M_x150.txt
