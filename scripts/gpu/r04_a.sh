#!/bin/bash
# round 4, call a: bisect of the 64-column clip instance at three waves per SIMD (variant libraries built here beforehand)
mkdir -p gpurun_out/r04a
timeout 900 python scripts/debug/clip64_bisect.py run > gpurun_out/r04a/clip64_bisect.txt 2> gpurun_out/r04a/err.txt
tail -60 gpurun_out/r04a/clip64_bisect.txt
tail -5 gpurun_out/r04a/err.txt
