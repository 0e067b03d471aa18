#!/bin/bash
# round 4, call af: clip_global by the plan's rule (staged form cramped: < 3 workgroups per CU or a shrunken tile): timings per read length, parity, the tools on long fixed-length reads
mkdir -p gpurun_out/r04af
SHAPES=100:20000000:1,150:20000000:7,152:10000000:7,176:10000000:1,176:10000000:7,188:10000000:1,200:10000000:1,252:8000000:1,300:6000000:1,300:6000000:7,1000:2000000:1 python scripts/debug/clip_global_vs_staged.py 2>&1 | grep -v amdgpu | tee gpurun_out/r04af/clip_global_vs_staged.txt | python -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('{'): continue
    d=json.loads(l); print(d['L'], d['stages'], 'staged', d['staged']['ms'], d['staged']['tile'], d['staged']['lds'], 'global', d['global']['ms'], d['global']['lds'], 'default', d['default']['ms'], d['default']['lds'])"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli.py -q -m gpu -x -k "over_the_batch or fuzz or adversarial or long_reads or configs_vs or first_n or long_fixed or cfg5 or full_size_cfg3" 2>&1 | tail -3
