#!/bin/bash
# round 4, call b: the clip-instance bisect with per-row dumps (variants built here; the emulator's dumps are made here too)
export BISECT_OUT=r04_bisect
mkdir -p gpurun_out/$BISECT_OUT
timeout 1200 python scripts/debug/clip64_bisect.py run > gpurun_out/$BISECT_OUT/run.txt 2> gpurun_out/$BISECT_OUT/err.txt
cat gpurun_out/$BISECT_OUT/run.txt; tail -5 gpurun_out/$BISECT_OUT/err.txt
