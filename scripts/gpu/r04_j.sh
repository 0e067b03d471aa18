#!/bin/bash
mkdir -p gpurun_out/r04j
timeout 600 python scripts/clip_roles_potential.py > gpurun_out/r04j/potential.txt 2>gpurun_out/r04j/err.txt
cat gpurun_out/r04j/potential.txt; tail -3 gpurun_out/r04j/err.txt
