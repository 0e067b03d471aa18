#!/usr/bin/env python3
"""Kernel time of every stage of the hot path that is not a BASELINE config of its own, on the cfg2-sized batch
(50 M x 150, seed 2), one GPU; not the driver's bench.  Algorithmic bytes = reads x (2 L + 4) + 2 x kept bytes (FASTQ in, res[], FASTQ out)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fastx_toolkit_amd import Engine, make_params  # noqa: E402

N, L = int(os.environ.get("READS", "50000000")), int(os.environ.get("LEN", "150"))
STAGES = [
    ("fastq_quality_trimmer -t 20 -l 30", dict(stages=2, qt_threshold=20, qt_min_len=30)),
    ("fastq_quality_filter -q 20 -p 80", dict(stages=4, qf_min_quality=20, qf_min_percent=80)),
    ("fastx_trimmer -f 5 -l 145", dict(stages=16, ft_first=5, ft_last=145)),
    ("fastx_reverse_complement", dict(stages=8)),
    ("fastq_masker -q 20", dict(stages=64, mask_min_quality=20)),
    ("fastx_artifacts_filter", dict(stages=128)),
    ("fastq_to_fasta (N filter)", dict(stages=256)),
]
eng = Engine(0)
eng.set_profiling(True)
b, q = eng.synth(2, 0, N, L, False)
outs = eng.alloc_outputs(N, L, compact=True, meta=False)
for name, pd in STAGES:
    P = make_params(**pd)
    ms = []
    for _ in range(5):
        r = eng.run(b, q, P, fixed_len=L, compact=True, meta=False, outputs=outs)
        ms.append(eng.last_kernel_ms())
    c = r.counters
    kept, kb = int(c[1]), int(c[2])
    alg = N * (2 * L + 4) + 2 * kb
    t = min(ms) * 1e-3
    print(json.dumps(dict(tool=name, kernel=eng.last_launch()["kernel"], ms_min=round(min(ms), 3), ms_med=round(sorted(ms)[2], 3), mreads_s=round(N / t / 1e6, 1),
                          alg_GB=round(alg / 1e9, 2), alg_TBs=round(alg / t / 1e12, 2), frac_hbm=round(alg / t / 8e12, 3), kept=kept, kept_bases=kb)), flush=True)
