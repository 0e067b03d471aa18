#!/bin/bash
# round 6, closing call after the reverse complement's dword-aligned instance: the evidence of r06_final.sh at HEAD, then the whole GPU tier, smoke() and the stage table
bash scripts/gpu/r06_final.sh
O=gpurun_out/r06bo; mkdir -p $O
timeout 3000 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -n 3 $O/pytest_gpu.txt | cut -c1-300
timeout 900 python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.txt 2>&1; tail -n 1 $O/smoke.txt | cut -c1-300
timeout 900 python scripts/bench_stages.py 2>&1 | grep "^{" > $O/stages_50M_x150.txt; cut -c1-200 $O/stages_50M_x150.txt
