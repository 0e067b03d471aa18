#!/usr/bin/env python3
"""CPU: a timed random campaign of the clip kernels' per-thread code (tests/emu) against the oracle -- every adapter class, strides 20..600, fixed and ragged
(with clip history), all flags, the DP over the staged tile / over the batch / as the plan picks.  `python scripts/fuzz_campaign.py <seed> <seconds>`; not a test (the test
tiers have fixed seeds), a tool for hunting what they miss."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import emu_py as emu
from helpers import assert_same, oracle_params, random_batch
from oracle import fxoracle_py as fo
rng = np.random.default_rng(int(sys.argv[1]))
adapters = [b"AGATCGGAAGAGC", b"CCTTAAGG", b"ACGT", b"TGGAATTCTCGGGTGCCAAGGAACTCCAGTCAC", b"AGATCGGAAGAGCACACGTCTGAACTCCAGTCACNNNNNNATCTCGTATGCCGTCTTCTGCTTG", b"ANNTCGNA", b"GATTACAGATTACAGA", b"A" * 17]
# round 5's buckets (56 and 80 columns, with and without N) and the widest ones, both ends of each
_r5 = np.random.default_rng(5)
for _n in (49, 52, 56, 57, 64, 65, 72, 80, 81, 99):
    _a = bytes(_r5.choice(list(b"ACGT"), size=_n).astype(np.uint8))
    adapters += [_a, _a[:_n // 2] + b"N" + _a[_n // 2 + 1:]]
# round 6: both ends of the new buckets (44, 52, 60, 72, 88), with and without N; bytes outside ACGTN (the pair table's lut): lower case, IUPAC codes with six
# distinct bytes (still the table) and with more (the general form); N in adapters of up to 16 bases (the register form serves them now)
_r6 = np.random.default_rng(6)
for _n in (41, 44, 45, 53, 60, 61, 73, 88, 89):
    _a = bytes(_r6.choice(list(b"ACGT"), size=_n).astype(np.uint8))
    adapters += [_a, _a[:_n // 3] + b"NN" + _a[_n // 3 + 2:]]
adapters += [b"agatcggaagagc", b"AGRYCGGAWGAGC", b"AGATCGGAAGAGCacacgtctgaactcc", b"ACGTRY" * 7, b"ACGTRYKMSWBDHVXZACGTACGT", b"NACGTNNACGTN", b"ANNNNNNNNNNNNNNT"]
t0 = time.time(); n = 0
while time.time() - t0 < float(sys.argv[2]):
    ad = adapters[int(rng.integers(0, len(adapters)))]
    stride = int(rng.choice([20, 36, 52, 64, 100, 152, 176, 188, 200, 252, 300, 400, 600]))
    nreads = int(rng.integers(1, 400))
    fixed = rng.random() < 0.6
    b, q, lens = random_batch(rng, nreads, stride, max(1, stride - int(rng.integers(0, stride))), stride, fixed, p_n=float(rng.choice([0.0, 0.02, 0.2])), adapter=ad)
    if rng.random() < 0.25:                                              # bytes of the adapter's own alphabet (and a few others) sprinkled over the reads
        hit = rng.random(b.shape) < 0.1
        b[hit] = rng.choice(np.frombuffer(bytes(sorted(set(ad))) + b"acgtnX", dtype=np.uint8), size=int(hit.sum()))
    stages = int(rng.choice([1, 7, 3, 5]))
    pd = dict(stages=stages, adapter=ad, clip_min_len=int(rng.integers(0, 25)), clip_flags=int(rng.integers(0, 16)), clip_min_adapter_len=int(rng.choice([0, 0, 3, 8])),
              qt_threshold=20, qt_min_len=int(rng.integers(0, 40)), qf_min_quality=int(rng.integers(0, 40)), qf_min_percent=int(rng.integers(0, 101)))
    for mode in ("0", "1", None):
        if mode is None: os.environ.pop("FXG_CLIP_GLOBAL", None)
        else: os.environ["FXG_CLIP_GLOBAL"] = mode
        if fixed:
            o = fo.run_pipeline(b, q, None, oracle_params(pd), fixed_len=stride); e = emu.run_pipeline(b, q, None, oracle_params(pd), fixed_len=stride)
        else:
            hs = emu.hist_new(); o = fo.run_pipeline(b, q, lens, oracle_params(pd)); e = emu.run_pipeline(b, q, lens, oracle_params(pd), hist=hs); emu.hist_free(hs)
        assert_same(o, e, "seed%s.n%d.ad%d.s%d.fixed%d.mode%s.%r" % (sys.argv[1], n, len(ad), stride, fixed, mode, pd))
    n += 1
print("seed", sys.argv[1], "cases", n, "ok")
