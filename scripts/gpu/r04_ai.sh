#!/bin/bash
# round 4, call ai: clip instances over the batch without the gather's granule table (LDS of long-read tiles): timings, whole GPU tier, closing evidence
mkdir -p gpurun_out/r04ai
SHAPES=300:6000000:1,600:3000000:1,1000:2000000:1 python scripts/debug/clip_global_vs_staged.py 2>&1 | grep -v amdgpu | tee gpurun_out/r04ai/clip_global_no_table.txt | python -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('{'): continue
    d=json.loads(l); print(d['L'], d['stages'], 'staged', d['staged']['ms'], d['staged']['tile'], d['staged']['lds'], 'global', d['global']['ms'], d['global']['lds'], 'default', d['default']['ms'], d['default']['lds'])"
ADAPTER=AGATCGGAAGAGCACACGTCTGAACTCCAGTCAC SHAPES=300:4000000:1,600:2000000:1 python scripts/debug/clip_global_vs_staged.py 2>&1 | grep -v amdgpu | tee -a gpurun_out/r04ai/clip_global_no_table.txt | python -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('{'): continue
    d=json.loads(l); print('truseq', d['L'], d['stages'], 'staged', d['staged']['ms'], d['staged']['lds'], 'global', d['global']['ms'], d['global']['lds'], 'default', d['default']['ms'], d['default']['lds'])"
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r04ai/pytest_gpu.txt 2>&1; grep -a "passed\|failed" gpurun_out/r04ai/pytest_gpu.txt | tail -2
bash scripts/gpu/r04_final.sh > gpurun_out/r04ai/final.log 2>&1; grep "traffic_over\|rc=" gpurun_out/r04ai/final.log | tail -12
