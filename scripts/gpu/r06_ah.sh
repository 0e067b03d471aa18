#!/bin/bash
# round 6, call ah: the whole GPU tier at HEAD (matrix included), smoke
O=gpurun_out/r06ah; mkdir -p $O
timeout 3000 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -n 6 $O/pytest_gpu.txt | cut -c1-300
timeout 600 python __graft_entry__.py smoke > $O/smoke.txt 2>&1; tail -n 2 $O/smoke.txt | cut -c1-300
