#!/bin/bash
# round 6, call c: float pair table in both passes (pass 2: exactly SPAN summary rows, the diagonal's summary step out of the table too)
O=gpurun_out/r06c; mkdir -p $O
LIBS=fastx_toolkit_amd/libfxg_v_noptab.so,fastx_toolkit_amd/libfxg_v_f32.so,fastx_toolkit_amd/libfxg.so timeout 900 python scripts/clip_ab.py > $O/clip_ab.txt 2>&1
cut -c1-400 $O/clip_ab.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "clip or fuzz or config or cfg5 or cfg3" > $O/pytest_clip.txt 2>&1; tail -n 5 $O/pytest_clip.txt
