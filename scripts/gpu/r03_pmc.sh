#!/bin/bash
# round-3 PMC call: FETCH_SIZE / WRITE_SIZE passes per config (separate passes, counters only) and the SQ counters of the clip kernel.
# usage: r03_pmc.sh <tag> "<cfg list>" ; results under gpurun_out/<tag>/ ; afterwards (here):  python scripts/pmc_traffic.py gpurun_out/<tag>/<cfg> <tag> <cfg>
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
TAG=${1:-r03pmc}; CFGS=${2:-"cfg2 cfg4 stats"}
export TMPDIR=/tmp
python -c "import bench; print(bench.csrc_sha16())" > /tmp/sha.txt
cd /tmp
for c in $CFGS; do
  O=$R/gpurun_out/$TAG/$c; mkdir -p $O; cp /tmp/sha.txt $O/csrc_sha16.txt
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/$ctr
    CFG=$c timeout 300 rocprofv3 --pmc $ctr -d $O/$ctr -o pmc --output-format csv -- python $R/scripts/pmc_run.py > $O/$ctr.log 2>&1
    echo "$c $ctr rc=$? $(grep '^cfg' $O/$ctr.log | cut -c1-160)"
    f=$(find $O/$ctr -name "*counter_collection.csv" | head -1); [ -n "$f" ] && [ "$f" != "$O/$ctr/pmc_counter_collection.csv" ] && mv $f $O/$ctr/pmc_counter_collection.csv
  done
done
