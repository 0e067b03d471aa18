#!/bin/bash
# round 5, call z: the closing state of the round -- the whole GPU tier, smoke, the default bench line
O=gpurun_out/r05z; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1
tail -6 $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 1500 python bench.py 2> $O/bench.err | grep "^{" > $O/bench_default.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05z/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic_over_algorithmic'), d['self_check']['matches_pinned'])
for k,v in d['configs'].items(): print(k, v.get('value'), v.get('ms_per_step'), (v.get('roofline') or {}).get('frac'), (v.get('self_check') or {}).get('matches_pinned'), v.get('error'))
e=d['e2e']; print({k:(v.get('mreads_s') if isinstance(v,dict) else v) for k,v in e.items() if k not in ('sharded_big','default_invocation')}); print({k:(v.get('mreads_s') if isinstance(v,dict) else v) for k,v in e['sharded_big'].items()})
PY
