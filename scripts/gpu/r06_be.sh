#!/bin/bash
# round 6, call be: what the alignment of the gather's source windows is worth (scripts/gather_alignment.py)
O=gpurun_out/r06be; mkdir -p $O
timeout 900 python scripts/gather_alignment.py 2>&1 | grep "^{" | tee $O/gather_alignment.txt | cut -c1-250
