#!/bin/bash
# round 5, call p: the whole GPU tier at the current tree (with the launch-bounds matrix), then smoke
O=gpurun_out/r05p; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.txt 2>&1
tail -15 $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
