#!/bin/bash
# round 4, call aa: SQ counters of the statistics kernel and of the clip kernel as they are now (one counter group per pass, counters only)
bash scripts/pmc_stats.sh 2>&1 | tail -4
python scripts/pmc_parse.py gpurun_out/pmc_stats/*/ 2>/dev/null | grep "quality_stats<\|quality_stats |" | grep -v fold > gpurun_out/pmc_stats/summary.txt; cat gpurun_out/pmc_stats/summary.txt | cut -c1-150
for c in cfg3 cfg5; do CFG=$c bash scripts/pmc_sq.sh r04_clip_pmc_$c scripts/pmc_clip.py 2>&1 | grep "fxg_kernel_tiles" | cut -c1-150; done
