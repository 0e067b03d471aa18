#!/bin/bash
# round 6, call bg: the reverse complement by row length (aligned / straddling / both), scripts/gather_alignment.py CASES=2
O=gpurun_out/r06bg; mkdir -p $O
CASES=2 timeout 900 python scripts/gather_alignment.py 2>&1 | grep "^{" | tee $O/gather_alignment_rev.txt | cut -c1-250
