#!/usr/bin/env python3
"""(kept reads, kept bases, checksum) of every rank's shard of the bench workloads (GPU box, not a test): rank g of a weak-scaling run owns reads
[g * R, (g + 1) * R) of the config's seed.  The tuples go into bench.EXPECTED; tests/test_gpu_parity.py::test_every_rank_shard_is_pinned checks each
of them against the oracle in windows before they are trusted.

    CFGS=cfg2,cfg5shard RANKS=8 python scripts/pin_shards.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from fastx_toolkit_amd import Engine, make_params  # noqa: E402

eng = Engine(0)
out = {}
for cfg in os.environ.get("CFGS", "cfg2,cfg5shard").split(","):
    c = bench.CONFIGS[cfg]
    R, L = c["reads"], c["L"]
    P = make_params(**c["params"])
    outs = eng.alloc_outputs(R, L, compact=True, meta=False)
    out[cfg] = []
    for g in range(int(os.environ.get("RANKS", "8"))):
        b, q = eng.synth(c["seed"], g * R, R, L, c["adapter"])
        r = eng.run(b, q, P, fixed_len=L, compact=True, meta=False, outputs=outs)
        t = (r.kept, r.kept_bytes, r.checksum())
        out[cfg].append(t)
        print(cfg, g, t, flush=True)
        del b, q, r
        torch.cuda.empty_cache()
    del outs
    torch.cuda.empty_cache()
print(json.dumps(out))
