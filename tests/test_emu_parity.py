"""CPU tier: the kernels' per-thread device code, run serially on the host (tests/emu), against the oracle.

The emulator executes the same __host__ __device__ phase bodies as the GPU kernels (bitmap build,
per-read decision incl. the forward-carried clipper DP, 16-byte chunk gather, reverse-complement);
only the wave-level scan / look-back is replaced by a serial prefix.  Bit-exact or it fails.
"""
import numpy as np

import emu_py as emu
from helpers import assert_same, fuzz_cases, oracle_params
from oracle import fxoracle_py as fo


def test_emulated_kernels_on_configs():
    cfgs = [
        ((2, 0, 20000, 150, False), dict(stages=6, qt_threshold=20, qt_min_len=30, qf_min_quality=20, qf_min_percent=80)),
        ((1, 0, 10000, 36, False), dict(stages=2, qt_threshold=20, qt_min_len=30)),
        ((2, 0, 20000, 150, False), dict(stages=24, ft_first=5, ft_last=145)),
        ((3, 0, 5000, 100, True), dict(stages=1, adapter=b"AGATCGGAAGAGC", clip_min_len=15, clip_flags=4)),
        ((5, 0, 5000, 150, True), dict(stages=7, adapter=b"AGATCGGAAGAGC", clip_min_len=15, clip_flags=4, qt_threshold=20,
                                       qt_min_len=30, qf_min_quality=20, qf_min_percent=80)),
    ]
    for gen, pd in cfgs:
        b, q = fo.synth_batch(*gen)
        p = oracle_params(pd)
        o = fo.run_pipeline(b, q, None, p)
        e = emu.run_pipeline(b, q, None, p)
        assert 0 < int(o["counters"][fo.C_KEPT]) <= gen[2]
        assert_same(o, e, str(pd))


def test_emulated_kernels_fuzz():
    kept_total = 0
    for name, b, q, lens, fl, pd in fuzz_cases(7, trials=40, clip_trials=32):
        p = oracle_params(pd)
        o = fo.run_pipeline(b, q, lens, p, fixed_len=fl)
        e = emu.run_pipeline(b, q, lens, p, fixed_len=fl)
        assert_same(o, e, name)
        kept_total += int(o["counters"][fo.C_KEPT])
    assert kept_total > 10000


def test_emulated_decision_only_and_bad_base():
    b, q = fo.synth_batch(9, 0, 3000, 50)
    p = oracle_params(dict(stages=6, qt_threshold=20, qt_min_len=10, qf_min_quality=15, qf_min_percent=70))
    o = fo.run_pipeline(b, q, None, p)
    e = emu.run_pipeline(b, q, None, p, compact=False)
    assert np.array_equal(o["res"], e["res"]) and np.array_equal(o["counters"][:13], e["counters"][:13])
    b2 = b.copy()
    b2[17, 3] = ord("X")
    e = emu.run_pipeline(b2, q, None, oracle_params(dict(stages=8)))
    assert int(e["counters"][15]) & 2          # fastx_reverse_complement.c:67-68
    e = emu.run_pipeline(b2, q, None, oracle_params(dict(stages=16, ft_first=2)))
    assert int(e["counters"][15]) == 0         # the fixed trimmer never looks at the alphabet


def test_emulated_clip_history_across_batches():
    """N3: the pre-pass that rebuilds the reference aligner's stale query tail, against the oracle's shared aligner."""
    from helpers import random_batch
    rng = np.random.default_rng(5)
    ad = b"AGATCGGAAGAGC"
    differs = 0
    for trial in range(24):
        stride = int(rng.choice([20, 36, 50, 75, 100, 151, 300]))
        al, hs = fo.aligner_new(), emu.hist_new()
        for batch in range(3):
            n = int(rng.integers(1, 900))
            st = stride if batch != 1 else max(5, stride // 2)          # a later batch narrower than the buffer
            b, q, lens = random_batch(rng, n, st, 1, st, False, adapter=ad)
            p = oracle_params(dict(stages=1 if trial % 2 else 7, adapter=ad if trial % 4 else ad * 3, clip_min_len=int(rng.integers(0, 10)),
                                   clip_flags=int(rng.integers(0, 16)), qt_threshold=20, qt_min_len=5, qf_min_quality=10, qf_min_percent=30))
            use_len = None if (batch == 2 and trial % 3 == 0) else lens  # a fixed-length batch after ragged ones still sees the tail
            fl = st if use_len is None else None
            o = fo.run_pipeline(b, q, use_len, p, fixed_len=fl, aligner=al)
            e = emu.run_pipeline(b, q, use_len, p, fixed_len=fl, hist=hs)
            assert_same(o, e, "hist.t%d.b%d" % (trial, batch))
            differs += int((emu.run_pipeline(b, q, use_len, p, fixed_len=fl)["res"] != o["res"]).sum())
        fo.aligner_free(al)
        emu.hist_free(hs)
    assert differs > 1000          # reads aligned on their own would have come out differently


def test_emulated_quality_stats_histogram():
    """fastx_quality_stats: the strip bodies of the reduction kernel against the oracle's per-cycle records, two batches of different width."""
    import ctypes as C
    from helpers import random_batch
    rng = np.random.default_rng(3)
    for trial in range(20):
        stride = int(rng.choice([1, 7, 16, 17, 36, 100, 150, 151, 300]))
        cols = stride + int(rng.integers(0, 20))
        hist, qs = None, fo.QStats()
        for batch in range(2):
            n = int(rng.integers(1, 3000))
            st = stride if batch == 0 else max(1, stride - int(rng.integers(0, min(stride, 10))))
            b, q, lens = random_batch(rng, n, st, 1, st, rng.random() < 0.4)
            if trial % 4 == 2:
                q = rng.integers(18, 126, size=q.shape, dtype=np.uint8)   # the whole legal range (-15 .. 92 at offset 33): also outside the LDS window
            use_q = q if trial % 5 else None                         # every fifth trial: FASTA-like batch, bin 0 counts
            hist = emu.run_quality_stats(b, use_q, lens, hist=hist, cols=cols)
            qs.add(b, use_q, lens, qoffset=33)
        if trial % 5:
            assert np.array_equal(hist, qs.device_layout(cols, 33)), (trial, stride)
        else:
            h = (C.c_int * fo.QS_RANGE)()
            for c in range(cols):
                for k in range(5):
                    assert int(hist[c, k, 0]) == fo.lib().fxo_qstats_hist(qs.h, c, k + 1, h) and int(hist[c, k, 1:].sum()) == 0
        qs.close()


def test_emulated_long_reads():
    """Reads up to the reference reader's line limit (24 999): tiles of a few reads, 157 column blocks in the statistics kernel."""
    from helpers import random_batch
    rng = np.random.default_rng(8)
    ad = b"AGATCGGAAGAGC"
    for stride in (24999, 1000):
        b, q, lens = random_batch(rng, 7, stride, stride // 2, stride, False, adapter=ad)
        for pd in (dict(stages=6, qt_threshold=20, qt_min_len=30, qf_min_quality=20, qf_min_percent=80), dict(stages=24, ft_first=5, ft_last=stride - 7),
                   dict(stages=8), dict(stages=64, mask_min_quality=20), dict(stages=128), dict(stages=1, adapter=ad, clip_min_len=15, clip_flags=4)):
            p = oracle_params(pd)
            hs = emu.hist_new() if pd["stages"] & 1 else None          # ragged clipper input: the oracle carries the aligner's history (N3)
            assert_same(fo.run_pipeline(b, q, lens, p), emu.run_pipeline(b, q, lens, p, hist=hs), "long.%d.%d" % (stride, pd["stages"]))
            if hs:
                emu.hist_free(hs)
        qs = fo.QStats()
        qs.add(b, q, lens, qoffset=33)
        assert np.array_equal(emu.run_quality_stats(b, q, lens), qs.device_layout(stride, 33))
        qs.close()


def test_emulated_clip_two_pass_adversarial():
    """The two-pass clipper (score pass + restart from a checkpoint, fxg_clip_two_pass) against the oracle's full matrix + traceback on
    inputs built to stress its bound on the best path's length: every adapter length 1..16 (all packed buckets of the two-pass form),
    low-complexity reads and adapters (long runs of ties), N-rich reads and adapters, adapters repeated along the read, reads shorter
    than the adapter, best cells in the first / last rows."""
    rng = np.random.default_rng(11)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    checked = 0
    for trial in range(96):
        alen = 1 + trial % 16
        kind = trial % 6
        if kind == 0:
            ad = bytes(rng.choice(acgt, size=alen))
        elif kind == 1:
            ad = bytes([int(rng.choice(acgt))]) * alen                          # homopolymer adapter
        elif kind == 2:
            ad = (b"AC" * alen)[:alen]
        elif kind == 3:
            ad = bytes(rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), size=alen))
        elif kind == 4:
            ad = (b"AGATCGGAAGAGCACAC")[:alen]
        else:
            ad = bytes(rng.choice(acgt[:2], size=alen))
        if ad.count(b"N") == alen:
            ad = b"A" + ad[1:]
        stride = int(rng.choice([3, 8, 17, 30, 64, 100, 150, 255]))
        n = int(rng.integers(50, 400))
        style = trial % 5
        if style == 0:
            b = rng.choice(acgt, size=(n, stride))
        elif style == 1:
            b = rng.choice(acgt[:2], size=(n, stride))                          # two-letter reads: ties everywhere
        elif style == 2:
            b = np.full((n, stride), ad[0], dtype=np.uint8)                     # homopolymer reads
            b[rng.random((n, stride)) < 0.05] = ord("C")
        elif style == 3:
            b = rng.choice(acgt, size=(n, stride))
            b[rng.random((n, stride)) < 0.25] = ord("N")
        else:
            reps = np.frombuffer((ad * (stride // len(ad) + 2))[:stride], dtype=np.uint8)
            b = np.tile(reps, (n, 1))
            b[rng.random((n, stride)) < 0.1] = rng.choice(acgt)
        b = np.ascontiguousarray(b)
        adv = np.frombuffer(ad, dtype=np.uint8)
        for i in range(0, n, 3):                                                 # plant (damaged) adapters, also at the very start / end
            pos = int(rng.choice([0, 1, max(0, stride - alen), max(0, stride - 2), int(rng.integers(0, stride))]))
            k = min(alen, stride - pos)
            a2 = adv.copy()
            if rng.random() < 0.5:
                a2[int(rng.integers(0, alen))] = rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8))
            b[i, pos:pos + k] = a2[:k]
            if rng.random() < 0.3 and pos > 1:                                   # an insertion / deletion right before it
                b[i, pos - 1] = b[i, pos]
        q = rng.integers(33, 75, size=(n, stride), dtype=np.uint8)
        for flags in (0, 4, int(rng.integers(0, 16))):
            pd = dict(stages=1, adapter=ad, clip_min_len=int(rng.integers(0, 12)), clip_min_adapter_len=int(rng.choice([0, 0, 2, 5])), clip_flags=flags)
            p = oracle_params(pd)
            o = fo.run_pipeline(b, q, None, p)
            e = emu.run_pipeline(b, q, None, p)
            assert_same(o, e, "clip2.t%d.a%s.s%d.f%d" % (trial, ad.decode(), stride, flags))
            checked += n
    assert checked > 40000
