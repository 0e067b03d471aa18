#!/bin/bash
# round-3 GPU call AD: the clip kernel per adapter length at HEAD (all instances in place), and the cfg5 tool chain end to end on a 32 M-read shard
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r03ad; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python scripts/clip_by_adapter_len.py 8 13 16 17 20 24 28 32 33 34 36 40 48 64 > $O/clip_100.txt 2> $O/err.txt; cut -c1-140 $O/clip_100.txt
L=150 timeout 600 python scripts/clip_by_adapter_len.py 13 20 24 34 48 64 > $O/clip_150.txt 2>> $O/err.txt; cut -c1-140 $O/clip_150.txt
L=250 READS=10000000 timeout 600 python scripts/clip_by_adapter_len.py 13 20 34 64 > $O/clip_250.txt 2>> $O/err.txt; cut -c1-140 $O/clip_250.txt
timeout 900 python bench.py --config cfg5shard --e2e --e2e-reads 32000000 --no-cpu-baseline --steps 3 --warmup 1 2> $O/e2e.err | grep "^{" > $O/cfg5shard_e2e_32m.json; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03ad/cfg5shard_e2e_32m.json").read().strip().splitlines()[-1])
print(json.dumps(d.get("e2e_ranks"))[:600])
PY
timeout 900 python bench.py --config cfg2 --e2e --e2e-reads 32000000 --no-cpu-baseline --no-e2e --steps 3 --warmup 1 2>> $O/e2e.err | grep "^{" > $O/cfg2_e2e_ranks_32m.json; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03ad/cfg2_e2e_ranks_32m.json").read().strip().splitlines()[-1])
print(json.dumps(d.get("e2e_ranks"))[:600])
PY
tail -2 $O/e2e.err
