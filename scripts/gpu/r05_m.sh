#!/bin/bash
# round 5, call m: the new GPU-tier tests (one file by strands, the rank path against the real RCCL, two ranks through the tool's rank mode), pipes, and the
# default bench invocation with every BASELINE config behind the headline
O=gpurun_out/r05m; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_cli.py -q -x -m gpu -k "one_output_file or sharded_run_and or lanes_devices" > $O/pytest_new.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "two_ranks or rccl" >> $O/pytest_new.txt 2>&1
timeout 300 python scripts/e2e_pipe.py > $O/e2e_pipe.txt 2>&1
( time timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time
tail -15 $O/pytest_new.txt; cat $O/e2e_pipe.txt; cat $O/bench_default.time; tail -c 400 $O/bench_default.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05m/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
for k,v in d.get('configs',{}).items():
    print(k, {kk:v.get(kk) for kk in ('value','ms_per_step','error','wall_s_incl_generation_and_cpu_baseline')}, (v.get('roofline') or {}).get('frac'), (v.get('roofline') or {}).get('useful_frac'), (v.get('self_check') or {}).get('matches_pinned'), (v.get('cpu_baseline') or {}).get('value'))
e=d.get('e2e',{})
print({k:(v.get('mreads_s') if isinstance(v,dict) else v) for k,v in e.items() if k not in ('sharded_big','default_invocation')})
print({k:(v.get('mreads_s') if isinstance(v,dict) else v) for k,v in e.get('sharded_big',{}).items()})
print(e.get('sharded_big',{}).get('one_file'))
PY
