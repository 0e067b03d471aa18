#!/bin/bash
# round 6, call ax: kernel time of the stages that are not BASELINE configs of their own (trimmer, filter, fixed trimmer, reverse complement, masker, artifacts filter,
# fastq_to_fasta's N filter) on the cfg2-sized batch
O=gpurun_out/r06ax; mkdir -p $O
timeout 900 python scripts/bench_stages.py 2>&1 | grep "^{" | tee $O/stages_50M_x150.txt | cut -c1-260
