#!/bin/bash
# round 4, call d: the 48-column N instance at four waves per SIMD on the wide fuzz (wrong on ragged input with history in call r04_c); also the
# GPU tier's new shard-vs-whole test
export CASES=wide-348 BISECT_OUT=r04_bisect_w4
mkdir -p gpurun_out/$BISECT_OUT gpurun_out/r04d
timeout 900 python scripts/debug/clip64_bisect.py run w4 w4_dbg1 w4_dbg2 w4_dbg3 > gpurun_out/$BISECT_OUT/run.txt 2> gpurun_out/$BISECT_OUT/err.txt
cat gpurun_out/$BISECT_OUT/run.txt; tail -5 gpurun_out/$BISECT_OUT/err.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "shards_reassemble or rccl_epilogue" > gpurun_out/r04d/pytest_shards.txt 2>&1; tail -15 gpurun_out/r04d/pytest_shards.txt
