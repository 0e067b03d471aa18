#!/usr/bin/env python3
"""Clip kernel time per adapter length (prefixes of the TruSeq adapter; the synthetic reads carry its first 13 bases).

    python scripts/clip_by_adapter_len.py [lengths...]      READS=20000000 L=100
One JSON line per length: which instance of fxg_kernel_tiles ran, ms, GCUPS over the full L x A matrix.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fastx_toolkit_amd import Engine, make_params  # noqa: E402

FULL = b"AGATCGGAAGAGCACACGTCTGAACTCCAGTCACATCACGATCTCGTATGCCGTCTTCTGCTTGAAAAAAAAAAGGGGGGGGGGCCCCCCCCCCTTTTTTTTTTACGT"      # 103 bases (the tool takes up to 99)
if os.environ.get("WITH_N"):                      # the index of the TruSeq adapter as NNNNNN (and one N in the prefix every length shares)
    FULL = b"AGATCGGAAGNGCACACGTCTGAACTCCAGTCACNNNNNNATCTCGTATGCCGTCTTCTGCTTG"
lens = [int(x) for x in sys.argv[1:]] or [8, 13, 16, 17, 20, 24, 28, 32, 33, 40, 48, 64]
n = int(os.environ.get("READS", "20000000"))
L = int(os.environ.get("L", "100"))
eng = Engine(0)
eng.set_profiling(True)
b, q = eng.synth(3, 0, n, L, True)
outs = eng.alloc_outputs(n, L, compact=True, meta=False)
for A in lens:
    P = make_params(stages=1, adapter=FULL[:A], clip_min_len=15, clip_flags=4)
    ms = []
    for _ in range(4):
        r = eng.run(b, q, P, fixed_len=L, compact=True, meta=False, outputs=outs)
        ms.append(eng.last_kernel_ms())
    t = min(ms) * 1e-3
    print(json.dumps(dict(adapter_len=A, kernel=eng.last_launch()["kernel"], grid=eng.last_launch()["grid"], ms_min=round(min(ms), 3),
                          mreads_s=round(n / t / 1e6, 1), gcups=round(n * L * A / t / 1e9, 1), kept=int(r.counters[1]), kept_bases=int(r.counters[2]))), flush=True)
