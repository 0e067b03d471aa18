#!/usr/bin/env python3
"""Debug (GPU box): the matrix worker's sequence of cases (tests/test_gpu_clip_matrix.py) through one library (FXG_LIB), reporting every
case that differs from the oracle instead of stopping at the first, re-running a failing case in the same context and in a fresh one."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "scripts", "debug"))
import numpy as np, torch
from helpers import adversarial_clip_cases, oracle_params
from oracle import fxoracle_py as fo
from fastx_toolkit_amd import Engine, make_params
import clip64_bisect as cb

def run(eng, b, q, lens, pd, hist):
    dl = torch.from_numpy(np.ascontiguousarray(lens).view(np.int16)).to(eng.device) if lens is not None else None
    eng.set_clip_history(hist)
    r = eng.run(eng.upload(b).view(b.shape), eng.upload(q).view(q.shape), make_params(**pd), lens=dl, fixed_len=None if lens is not None else b.shape[1], compact=True).to_host()
    eng.set_clip_history(False)
    return r, eng.last_launch()

eng = Engine(0)
skip_adv = os.environ.get("SKIP_ADVERSARIAL")
n = 0
if not skip_adv:
    for long_adapters in (False, True):
        for name, b, q, pd in adversarial_clip_cases(long_adapters):
            r, _ = run(eng, b, q, None, pd, False); n += 1
            o = fo.run_pipeline(b, q, None, oracle_params(pd))
            if not np.array_equal(o["res"], r["res"]):
                print("adversarial", name[:40], "differs in", int((o["res"] != r["res"]).sum()), "reads")
print("adversarial cases run:", n, flush=True)
for name, b, q, lens, pd, hist in cb.wide_cases(os.environ.get("ONLY", "").split(",") if os.environ.get("ONLY") else None):
    o = fo.run_pipeline(b, q, lens, oracle_params(pd))
    if os.environ.get("SCRAMBLE"):                           # leave unrelated values in the waves' scratch: the same instance on other data first
        rs = np.random.default_rng(5)
        b2 = rs.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), size=b.shape)
        l2 = rs.integers(1, b.shape[1] + 1, size=b.shape[0]).astype(np.uint16) if lens is not None else None
        for _ in range(int(os.environ["SCRAMBLE"])):
            run(eng, np.ascontiguousarray(b2), q, l2, pd, hist)
    if os.environ.get("FRESH"):                              # a new context per case: new checkpoint scratch, history workspace and status arrays
        eng.close(); eng = Engine(0)
    r, ll = run(eng, b, q, lens, pd, hist)
    d = np.nonzero(o["res"] != r["res"])[0]
    if len(d):
        r2, _ = run(eng, b, q, lens, pd, hist)
        e2 = Engine(0); r3, _ = run(e2, b, q, lens, pd, hist); e2.close()
        print(name, ll["kernel"].split()[0], "grid", ll["grid"], "tile", ll["tile_reads"], "n", b.shape[0], ": %d reads differ; again in the same context: %d; in a fresh context: %d" % (
            len(d), int((o["res"] != r2["res"]).sum()), int((o["res"] != r3["res"]).sum())))
        for i in d[:10]:
            print("    read %d len %d oracle %06x engine %06x" % (i, lens[i] if lens is not None else b.shape[1], o["res"][i], r["res"][i]))
    else:
        print(name, ll["kernel"].split()[0], "ok", flush=True)
