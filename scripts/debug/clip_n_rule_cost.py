import sys, os, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from fastx_toolkit_amd import Engine, make_params
eng = Engine(0)
R, L = 20_000_000, 100
b, q = eng.synth(3, 0, R, L, True)
outs = eng.alloc_outputs(R, L, compact=True, meta=False)
eng.set_profiling(True)
for flags in (4, 0):
    P = make_params(stages=1, adapter=b"AGATCGGAAGAGC", clip_min_len=15, clip_flags=flags)
    ms = []
    for _ in range(4):
        r = eng.run(b, q, P, fixed_len=L, compact=True, meta=False, outputs=outs); ms.append(eng.last_kernel_ms())
    print(json.dumps(dict(clip_flags=flags, ms_min=round(min(ms), 3), kept=int(r.counters[1]))))
