#!/bin/bash
# round 4, call ag: clip_global for the 17..99-column (checkpoint) instances: TruSeq adapters on long reads, staged / over the batch / the plan's pick; parity of both forms
mkdir -p gpurun_out/r04ag
for ad in AGATCGGAAGAGCACACGTCTGAACTCCAGTCAC AGATCGGAAGAGCACACGTCTGAACTCCAGTCACNNNNNNATCTCGTATGCCGTCTTCTGCTTG; do
ADAPTER=$ad SHAPES=100:10000000:1,152:10000000:1,152:10000000:7,200:6000000:1,252:6000000:1,300:4000000:1,300:4000000:7,600:2000000:1 python scripts/debug/clip_global_vs_staged.py 2>&1 | grep -v amdgpu | tee -a gpurun_out/r04ag/clip_global_vs_staged_long_adapters.txt | python -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('{'): continue
    d=json.loads(l); print(len('$ad'), d['L'], d['stages'], d['staged']['kernel'], 'staged', d['staged']['ms'], d['staged']['tile'], d['staged']['lds'], 'global', d['global']['ms'], d['global']['lds'], 'default', d['default']['ms'], d['default']['lds'])"
done
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_clip_matrix.py -q -m gpu -x -k "over_the_batch or adversarial or long_reads or matrix" 2>&1 | tail -3
