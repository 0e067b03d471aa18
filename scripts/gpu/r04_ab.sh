#!/bin/bash
# round 4, call ab: the statistics kernel with the next trip's rows requested before the current ones are added (software pipeline), UNROLL 1 / 2 / 3
mkdir -p gpurun_out/r04ab
(echo "ship $(python bench.py --config stats --no-cpu-baseline --steps 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms_avg'], d['roofline']['frac'])")"
python scripts/r04_variants.py run qs_pipe qs_pipe_u1 qs_pipe_u3) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04ab/stats_pipeline.txt
FXG_LIB=$PWD/fastx_toolkit_amd/libfxg_x_qs_pipe.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli.py -q -m gpu -k "stats" 2>&1 | tail -3
