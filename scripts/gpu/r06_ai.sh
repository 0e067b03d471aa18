#!/bin/bash
# round 6, call ai: smoke() in the three ways a driver may call it (after build() in the same process, on its own, as the script's main)
O=gpurun_out/r06ai; mkdir -p $O
( timeout 900 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -n 2 | cut -c1-200
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 1 | cut -c1-200
  timeout 900 python __graft_entry__.py smoke 2>&1 | tail -n 1 | cut -c1-200 ) | tee $O/smoke_three_ways.txt
