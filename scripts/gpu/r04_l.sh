#!/bin/bash
mkdir -p gpurun_out/r04l
timeout 1200 python scripts/r04_variants.py run base ticket_after_dp ticket_last_moment > gpurun_out/r04l/variants2.txt 2> gpurun_out/r04l/err.txt
cat gpurun_out/r04l/variants2.txt; tail -3 gpurun_out/r04l/err.txt
for v in base ticket_after_dp ticket_last_moment; do FXG_LIB=fastx_toolkit_amd/libfxg_x_$v.so python bench.py --config cfg4 --no-cpu-baseline --steps 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v cfg4', d['roofline']['kernel_ms_avg'])"; done
