#!/bin/bash
# round 5, last call: the bench line's e2e leg (the tools through bench.py's own code, pipes included) on the final host build
O=gpurun_out/r05last; mkdir -p $O
timeout 160 python bench.py --headline-only --no-cpu-baseline --steps 5 --warmup 1 2> $O/bench.err | grep "^{" > $O/bench_e2e.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05last/bench_e2e.json').read().strip().splitlines()[-1])
e=d['e2e']; print(d['value'], {k:(v.get('mreads_s') if isinstance(v,dict) else v) for k,v in e.items() if k not in ('sharded_big','default_invocation')}); print({k:(v.get('mreads_s') if isinstance(v,dict) else v) for k,v in e['sharded_big'].items()})
PY
tail -n 3 $O/bench.err
