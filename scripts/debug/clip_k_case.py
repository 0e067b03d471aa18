#!/usr/bin/env python3
"""Debug: adversarial clip cases on the GPU, two-pass K vs one-pass K vs oracle; prints the reads that differ."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import adversarial_clip_cases, oracle_params
from oracle import fxoracle_py as fo
from fastx_toolkit_amd import Engine, make_params
eng = Engine(0)
def run(b, q, pd):
    r = eng.run(eng.upload(b).view(b.shape), eng.upload(q).view(q.shape), make_params(**pd), fixed_len=b.shape[1], compact=True)
    return r.to_host()
bad = 0
for name, b, q, pd in adversarial_clip_cases(True):
    if b"N" in pd["adapter"]:
        continue
    o = fo.run_pipeline(b, q, None, oracle_params(pd))
    os.environ.pop("FXG_CLIP_K_ONE_PASS", None)
    e2 = run(b, q, pd); k2 = eng.last_launch()["kernel"]
    os.environ["FXG_CLIP_K_ONE_PASS"] = "1"
    e1 = run(b, q, pd); k1 = eng.last_launch()["kernel"]
    d2 = np.nonzero(o["res"] != e2["res"])[0]; d1 = np.nonzero(o["res"] != e1["res"])[0]
    if len(d2) or len(d1):
        bad += 1
        print(name[:20], "A", len(pd["adapter"]), "stride", b.shape[1], "n", b.shape[0], k2.split()[0], "two-pass diffs", len(d2), "one-pass diffs", len(d1), k1.split()[0])
        for i in d2[:4]:
            print("   read", i, "oracle %x two %x one %x" % (o["res"][i], e2["res"][i], e1["res"][i]), bytes(b[i]).decode())
        print("   adapter", pd["adapter"].decode(), pd)
print("cases with differences:", bad)
