#!/usr/bin/env python3
"""GPU box: a timed random campaign of the product's kernels through the C-ABI against the oracle (`python scripts/fuzz_campaign_gpu.py <seed> <seconds>`).
Not a test (the GPU tier has fixed seeds), a tool for hunting what it misses.  Each case is one random request:
  * clip [+ quality trim] [+ quality filter]: the adapter classes of scripts/fuzz_campaign.py (every bucket's ends, with and without N, odd alphabets), strides 20..600,
    1..3000 reads, fixed and ragged (with clip history), all flags, the DP over the staged tile / over the batch / as the plan picks;
  * quality trim + filter alone at strides 1..310 (both quality kernels: FXG_ROWS 0 / 1), fixed trim, reverse-complement, masker, artifacts filter;
  * quality statistics of the same batch (fixed-length batches take the round-robin loop, ragged ones the tested loop)."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
from helpers import assert_same, oracle_params, random_batch
from oracle import fxoracle_py as fo
from fastx_toolkit_amd import Engine, make_params

rng = np.random.default_rng(int(sys.argv[1]))
adapters = [b"AGATCGGAAGAGC", b"CCTTAAGG", b"ACGT", b"TGGAATTCTCGGGTGCCAAGGAACTCCAGTCAC", b"AGATCGGAAGAGCACACGTCTGAACTCCAGTCACNNNNNNATCTCGTATGCCGTCTTCTGCTTG", b"ANNTCGNA", b"GATTACAGATTACAGA", b"A" * 17]
_r = np.random.default_rng(56)
for _n in (17, 20, 21, 24, 33, 36, 37, 41, 44, 45, 49, 52, 53, 56, 57, 60, 61, 64, 65, 72, 73, 80, 81, 88, 89, 99):
    _a = bytes(_r.choice(list(b"ACGT"), size=_n).astype(np.uint8))
    adapters += [_a, _a[:_n // 3] + b"N" + _a[_n // 3 + 1:]]
adapters += [b"agatcggaagagc", b"AGRYCGGAWGAGC", b"AGATCGGAAGAGCacacgtctgaactcc", b"ACGTRY" * 7, b"ACGTRYKMSWBDHVXZACGTACGT", b"NACGTNNACGTN", b"ANNNNNNNNNNNNNNT"]


def engine_params(pd):
    return make_params(**{k: v for k, v in pd.items()})


def run(eng, b, q, lens, pd, fixed_len=None):
    dl = torch.from_numpy(np.ascontiguousarray(lens).view(np.int16)).to(eng.device) if lens is not None else None
    return eng.run(eng.upload(b).view(b.shape), eng.upload(q).view(q.shape), engine_params(pd), lens=dl, fixed_len=fixed_len, compact=True).to_host()


t0 = time.time(); n = 0; kinds = {}
eng = Engine(0)
while time.time() - t0 < float(sys.argv[2]):
    kind = str(rng.choice(["clip", "clip", "clip", "quality", "other"]))
    fixed = rng.random() < 0.6
    if kind == "clip":
        ad = adapters[int(rng.integers(0, len(adapters)))]
        stride = int(rng.choice([20, 36, 52, 64, 100, 150, 152, 176, 188, 200, 252, 300, 400, 600]))
        nreads = int(rng.integers(1, 3000 if len(ad) <= 36 else 700))
        b, q, lens = random_batch(rng, nreads, stride, max(1, stride - int(rng.integers(0, stride))), stride, fixed, p_n=float(rng.choice([0.0, 0.02, 0.2])), adapter=ad)
        if rng.random() < 0.25:
            hit = rng.random(b.shape) < 0.1
            b[hit] = rng.choice(np.frombuffer(bytes(sorted(set(ad))) + b"acgtnX", dtype=np.uint8), size=int(hit.sum()))
        pd = dict(stages=int(rng.choice([1, 7, 3, 5])), adapter=ad, clip_min_len=int(rng.integers(0, 25)), clip_flags=int(rng.integers(0, 16)),
                  clip_min_adapter_len=int(rng.choice([0, 0, 3, 8])), qt_threshold=20, qt_min_len=int(rng.integers(0, 40)), qf_min_quality=int(rng.integers(0, 40)),
                  qf_min_percent=int(rng.integers(0, 101)))
        mode = rng.choice(["0", "1", ""])
        if mode: os.environ["FXG_CLIP_GLOBAL"] = str(mode)
        else: os.environ.pop("FXG_CLIP_GLOBAL", None)
    else:
        stride = int(rng.integers(1, 311))
        nreads = int(rng.integers(1, 20000))
        b, q, lens = random_batch(rng, nreads, stride, 1, stride, fixed)
        if kind == "quality":
            pd = dict(stages=int(rng.choice([2, 4, 6])), qt_threshold=int(rng.integers(0, 45)), qt_min_len=int(rng.integers(0, stride + 2)), qf_min_quality=int(rng.integers(0, 45)),
                      qf_min_percent=int(rng.integers(0, 101)))
            os.environ["FXG_ROWS"] = str(rng.choice(["0", "1"]))
        else:
            pd = [dict(stages=16, ft_first=int(rng.integers(1, stride + 2)), ft_last=int(rng.integers(0, stride + 2))), dict(stages=8),
                  dict(stages=24, ft_first=int(rng.integers(1, stride + 1)), ft_last=int(rng.integers(0, stride + 2))), dict(stages=32, ft_trim_end=int(rng.integers(0, stride + 1)), ft_min_len=int(rng.integers(0, 20))),
                  dict(stages=64, mask_min_quality=int(rng.integers(0, 45))), dict(stages=128), dict(stages=256, nf_keep_n=int(rng.integers(0, 2)))][int(rng.integers(0, 7))]
    tag = "seed%s.case%d.%s.stride%d.n%d.fixed%d.%r.%s" % (sys.argv[1], n, kind, stride, nreads, fixed, pd, {k: os.environ.get(k) for k in ("FXG_CLIP_GLOBAL", "FXG_ROWS")})
    if fixed:
        o = fo.run_pipeline(b, q, None, oracle_params(pd), fixed_len=stride)
        e = run(eng, b, q, None, pd, fixed_len=stride)
    else:
        eng.set_clip_history(bool(pd["stages"] & 1))
        o = fo.run_pipeline(b, q, lens, oracle_params(pd))
        e = run(eng, b, q, lens, pd)
        eng.set_clip_history(False)
    assert_same(o, e, tag)
    if n % 3 == 0:                                                       # the statistics kernel on the same batch
        qs = fo.QStats(); qs.add(b, q, None if fixed else lens, qoffset=33)
        dl = None if fixed else torch.from_numpy(np.ascontiguousarray(lens).view(np.int16)).to(eng.device)
        h = eng.quality_stats(eng.upload(b).view(b.shape), eng.upload(q).view(q.shape), lens=dl, fixed_len=stride if fixed else None)
        assert np.array_equal(h.cpu().numpy().astype(np.uint64), qs.device_layout(stride, 33)), "stats " + tag
        qs.close()
    os.environ.pop("FXG_CLIP_GLOBAL", None); os.environ.pop("FXG_ROWS", None)
    kinds[kind] = kinds.get(kind, 0) + 1
    n += 1
print("seed", sys.argv[1], "cases", n, kinds, "all equal to the oracle")
