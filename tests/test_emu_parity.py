"""CPU tier: the kernels' per-thread device code, run serially on the host (tests/emu), against the oracle.

The emulator executes the same __host__ __device__ phase bodies as the GPU kernels (bitmap build,
per-read decision incl. the forward-carried clipper DP, 16-byte chunk gather, reverse-complement);
only the wave-level scan / look-back is replaced by a serial prefix.  Bit-exact or it fails.
"""
import numpy as np
import pytest

import emu_py as emu
from helpers import adversarial_clip_cases, assert_same, first_n_cases, fuzz_cases, odd_alphabet_clip_cases, oracle_params
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


def test_emulated_clip_odd_alphabets():
    """Bytes outside ACGTN in the adapter and in the reads: every byte value goes through the pair table's lut like the reference's `==`."""
    clipped = 0
    for name, b, q, fl, pd in odd_alphabet_clip_cases():
        p = oracle_params(pd)
        o = fo.run_pipeline(b, q, None, p, fixed_len=fl)
        e = emu.run_pipeline(b, q, None, p, fixed_len=fl)
        assert_same(o, e, name)
        clipped += int(((o["res"] >> 21) & 1).sum())
    assert clipped > 500


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
    # the base census (fastx_artifacts_filter.c:70-95, fastq_to_fasta): upper-case A C G T N only, at every byte position of a dword and in the read's last,
    # partial dword; a clean batch raises nothing
    for st in (128, 256):
        assert int(emu.run_pipeline(b, q, None, oracle_params(dict(stages=st)))["counters"][15]) == 0
        for pos, ch in ((0, "X"), (1, "a"), (2, "@"), (3, "n"), (48, "."), (49, "c"), (23, "\x00"), (37, "\xff")):
            b3 = b.copy()
            b3[41, pos] = ord(ch)
            assert int(emu.run_pipeline(b3, q, None, oracle_params(dict(stages=st)))["counters"][15]) & 2, (st, pos, ch)


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


def test_emulated_clip_first_n_rule():
    """the -n rule's scan of the read for its first N (four bases at a time where the rows start on a dword boundary)"""
    dropped = 0
    for name, b, q, lens, fl, pd in first_n_cases():
        hs = emu.hist_new()          # ragged reads: the reference's aligner carries its query buffer from read to read (N3), and so does the oracle
        o = fo.run_pipeline(b, q, lens, oracle_params(pd), fixed_len=fl)
        e = emu.run_pipeline(b, q, lens, oracle_params(pd), fixed_len=fl, hist=hs)
        emu.hist_free(hs)
        assert_same(o, e, name)
        dropped += int(len(b) - o["counters"][1])
    assert dropped > 1000


@pytest.mark.parametrize("mode", ["0", "1"])
def test_emulated_clip_over_the_batch_and_over_the_staged_tile(monkeypatch, mode):
    """fxg_plan.h clip_global: the register two-pass clip instances run their DP over a staged tile in LDS (0) or straight over the batch through a
    two-dword window (1); the default picks by row length and stages.  Both forms, forced, on everything the clip tests have for fixed-length
    batches and for ragged ones without history: configs 3 and 5, the clip fuzz, the adversarial corpus of the short adapters, the N-rule cases."""
    monkeypatch.setenv("FXG_CLIP_GLOBAL", mode)
    for args, pd in (((3, 0, 3000, 100, True), dict(stages=1, adapter=b"AGATCGGAAGAGC", clip_min_len=15, clip_flags=4)),
                     ((5, 0, 3000, 150, True), dict(stages=7, adapter=b"AGATCGGAAGAGC", clip_min_len=15, clip_flags=4, qt_threshold=20, qt_min_len=30, qf_min_quality=20, qf_min_percent=80)),
                     ((5, 0, 1500, 300, True), dict(stages=1, adapter=b"AGATCGGAAGAGC", clip_min_len=15, clip_flags=0)),
                     ((8, 0, 1200, 252, True), dict(stages=1, adapter=b"AGATCGGAAGAGCACACGTCTGAACTCCAGTCAC", clip_min_len=15, clip_flags=0)),
                     ((9, 0, 800, 300, True), dict(stages=7, adapter=b"AGATCGGAAGAGCACACGTCTGAACTCCAGTCACNNNNNNATCTCGTATGCCGTCTTCTGCTTG", clip_min_len=15, clip_flags=4,
                                                   qt_threshold=20, qt_min_len=30, qf_min_quality=20, qf_min_percent=80))):
        b, q = fo.synth_batch(*args)
        assert_same(fo.run_pipeline(b, q, None, oracle_params(pd)), emu.run_pipeline(b, q, None, oracle_params(pd)), "cfg%d.global%s" % (args[0], mode))
    n = 0
    for name, b, q, lens, fl, pd in fuzz_cases(23, trials=0, clip_trials=24):
        assert_same(fo.run_pipeline(b, q, lens, oracle_params(pd), fixed_len=fl), emu.run_pipeline(b, q, lens, oracle_params(pd), fixed_len=fl), name + ".global" + mode)
        n += 1
    for long_adapters in (False, True):                      # 1..16 columns: the register form; 17..99: the checkpoint form
        for name, b, q, pd in adversarial_clip_cases(long_adapters):
            assert_same(fo.run_pipeline(b, q, None, oracle_params(pd)), emu.run_pipeline(b, q, None, oracle_params(pd)), name + ".global" + mode)
            n += 1
    for name, b, q, lens, fl, pd in first_n_cases():
        if lens is None:
            assert_same(fo.run_pipeline(b, q, None, oracle_params(pd), fixed_len=fl), emu.run_pipeline(b, q, None, oracle_params(pd), fixed_len=fl), name + ".global" + mode)
            n += 1
    assert n > 60


def test_clip_global_window_clamp_on_huge_batches():
    """The over-the-batch clip form clamps its two-dword window to the array's last dword.  The number of dwords behind a row is a 64-bit count:
    a batch of 8 GiB and more (30 M reads at a 300-byte stride) used to wrap it into a negative int for its early rows, and every window load
    then went to a negative index.  The helper the kernels call, on totals no test could allocate."""
    import ctypes as C
    f = emu.lib().fxg_emu_gl_last_dword
    f.argtypes = [C.c_uint64, C.c_uint64]
    f.restype = C.c_int
    assert f(1200, 0) == 299 and f(1200, 900) == 74 and f(1200, 1196) == 0 and f(1200, 1200) == 0 and f(4, 0) == 0
    for total in (8 << 30, (8 << 30) + 4, 30_000_000 * 300, 1 << 40, (1 << 64) - 4):
        assert f(total, 0) == 0x7FFFFFFF, total                 # far more than an int of dwords left: "no clamp within reach", never negative
        assert f(total, total - 400) == 99
        assert f(total, total - 4) == 0
        assert f(total, total - (4 << 31)) == 0x7FFFFFFF - 0    # exactly 2^31 dwords left -> last index 2^31 - 1
        if total > (4 << 31) + 4:
            assert f(total, total - (4 << 31) - 4) == 0x7FFFFFFF


def test_clip_global_selection(monkeypatch):
    """fxg_make_plan (host logic shared with the engine): when the clip DP runs over the batch instead of a staged tile -- where the staged form would
    shrink its tile or keep fewer than three workgroups on a CU, rows on dword boundaries, no clip history."""
    monkeypatch.delenv("FXG_CLIP_GLOBAL", raising=False)
    ad, truseq = b"AGATCGGAAGAGC", b"AGATCGGAAGAGCACACGTCTGAACTCCAGTCAC"
    q7 = dict(qt_threshold=20, qt_min_len=30, qf_min_quality=20, qf_min_percent=80)
    for stride, stages, adapter, want, tile in ((100, 1, ad, False, 256), (150, 7, ad, False, 256), (152, 7, ad, False, 256), (176, 1, ad, False, 256), (176, 7, ad, True, 256),
                                                (188, 1, ad, True, 256), (250, 1, ad, False, 128), (252, 1, ad, True, 256), (300, 7, ad, True, 256), (1000, 1, ad, True, 256),
                                                (100, 1, truseq, False, 256), (152, 1, truseq, False, 256), (200, 1, truseq, True, 256), (300, 1, truseq, True, 256)):
        b, q = fo.synth_batch(3, 0, 300, stride, True)
        pd = dict(stages=stages, adapter=adapter, clip_min_len=5, clip_flags=4, **(q7 if stages == 7 else {}))
        emu.run_pipeline(b, q, None, oracle_params(pd))
        assert emu.last_plan_clip_global() == (want, tile), (stride, stages, len(adapter), emu.last_plan_clip_global())
    # a run with clip history keeps the staged form whatever the rows look like
    b, q = fo.synth_batch(3, 0, 300, 300, True)
    hs = emu.hist_new()
    emu.run_pipeline(b, q, np.full(300, 300, dtype=np.uint16), oracle_params(dict(stages=1, adapter=ad, clip_min_len=5, clip_flags=4)), hist=hs)
    emu.hist_free(hs)
    assert emu.last_plan_clip_global()[0] is False


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


def test_emulated_quality_stats_piece_form(monkeypatch):
    """The statistics kernel's piece form (dense rows of length 16 .. 160: lane p takes the p-th aligned 16-byte piece of every trip of whole reads): every
    eligible length class (gcd(L, 16) = 1, 2, 4, 8, 16; odd lengths keep block and counter half per byte), reads behind the last whole trip, lower-case and foreign letters, qualities outside the LDS window, the
    wrap guard of the 16-bit counters -- against the oracle, and equal to the row-strip form on the same batch."""
    from helpers import random_batch
    rng = np.random.default_rng(17)
    for trial, L in enumerate([16, 18, 20, 24, 36, 50, 76, 100, 126, 144, 150, 158, 160, 150, 100, 36, 17, 51, 75, 101, 151, 159, 33, 151]):
        n = int(rng.integers(1, 4000)) if trial % 3 else int(rng.integers(900, 1100))
        b, q, _ = random_batch(rng, n, L, L, L, True)
        if trial % 2:
            hit = rng.random(b.shape) < 0.01
            b[hit] = rng.choice(np.frombuffer(b"acgtn@XR.", dtype=np.uint8), size=int(hit.sum()))
            hit = rng.random(q.shape) < 0.01
            q[hit] = rng.integers(18, 126, size=int(hit.sum()), dtype=np.uint8)
        if trial in (13, 14, 15, 22, 23):
            monkeypatch.setenv("FXG_EMU_QS_FLUSH", "1")
        if trial % 4 == 3:
            monkeypatch.setenv("FXG_EMU_QS_NOGRID", "1")           # a launch whose workgroups are not a multiple of eight: cuts at whole trips
        cols = L + int(rng.integers(0, 5))
        h = emu.run_quality_stats(b, q, None, hist=None, cols=cols)
        monkeypatch.setenv("FXG_EMU_QS_ROWS", "1")
        h_rows = emu.run_quality_stats(b, q, None, hist=None, cols=cols)
        monkeypatch.delenv("FXG_EMU_QS_ROWS")
        monkeypatch.delenv("FXG_EMU_QS_FLUSH", raising=False)
        monkeypatch.delenv("FXG_EMU_QS_NOGRID", raising=False)
        qs = fo.QStats()
        qs.add(b, q, None, qoffset=33)
        assert np.array_equal(h, qs.device_layout(cols, 33)), (trial, L, n)
        assert np.array_equal(h, h_rows), (trial, L, n)
        qs.close()
    assert emu.lib().fxg_emu_quality_stats_piece_trips() > 0           # the form under test did run,
    assert emu.lib().fxg_emu_quality_stats_piece_moved() > 0           # with cuts between trips that were moved onto the 128-byte grid


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


@pytest.mark.parametrize("form", ["short", "long", "long-one-pass", "short-general", "long-general"])
def test_emulated_clip_two_pass_adversarial(form, monkeypatch):
    """The two-pass clipper (score pass + restart from a checkpoint, fxg_clip_two_pass) and, for adapters of 17..99 bases, its
    counterpart with the checkpoints in scratch (fxg_clip_two_pass_k; reads beyond 255 bases included) and the one-pass in-place form
    (fxg_clip_rows_k: short reads, or everywhere with FXG_CLIP_K_ONE_PASS) against the oracle's full matrix + traceback on
    helpers.adversarial_clip_cases.  Adapters that contain N run the same forms with the neutral-column selects; FXG_NO_PACKED_CLIP
    sends everything through the general two-word form."""
    checked = 0
    if form == "long-one-pass":
        monkeypatch.setenv("FXG_CLIP_K_ONE_PASS", "1")
    if form.endswith("general"):                       # the two-word form every adapter took before the packed ones: still the fallback
        monkeypatch.setenv("FXG_NO_PACKED_CLIP", "1")
    for name, b, q, pd in adversarial_clip_cases(form.startswith("long")):
        p = oracle_params(pd)
        assert_same(fo.run_pipeline(b, q, None, p), emu.run_pipeline(b, q, None, p), name)
        checked += b.shape[0]
    assert checked > 40000


def test_clip_instance_selection(monkeypatch):
    """fxg_make_plan (host logic shared with the engine): which clip instance a request gets.  A request that silently fell back to the
    general form (positive codes) would still be right, only 5-50 times slower -- so the choice itself is pinned here."""
    rng = np.random.default_rng(3)
    full = b"AGATCGGAAGAGCACACGTCTGAACTCCAGTCACATCACGATCTCGTATGCCGTCTTCTGCTTGAAAAAAAAAAGGGGGGGGGGCCCCCCCCCCTTTTTTTTT"
    with_n = b"AGATCGGAAGNGCACACGTCTGAACTCCAGTCACNNNNNNATCTCGTATGCCGTCTTCTGCTTGAAAAAAAAAAGGGGGGGGGGCCCCCCCCCCTTTTTTTTT"
    cases = [  # adapter, stride, instance, two passes with checkpoints in scratch
        (full[:4], 100, -4, False), (full[:7], 100, -8, False), (full[:13], 100, -13, False), (full[:16], 255, -16, False),
        (full[:13], 300, -13, False), (full[:13], 1000, -13, False),                      # the register form takes reads of any length
        (full[:17], 100, -20, True), (full[:17], 50, -20, True), (full[:17], 25, -20, False), (full[:24], 100, -24, True), (full[:32], 100, -32, True),
        (full[:33], 100, -36, True), (full[:34], 150, -36, True), (full[:34], 60, -36, True), (full[:34], 50, -36, False), (full[:40], 100, -40, True),
        (full[:41], 150, -44, True), (full[:44], 150, -44, True), (full[:45], 150, -48, True), (full[:48], 150, -48, True), (full[:49], 150, -52, True), (full[:52], 150, -52, True),
        (full[:53], 150, -56, True), (full[:56], 150, -56, True), (full[:57], 150, -60, True), (full[:61], 150, -64, True), (full[:64], 150, -64, True), (full[:64], 100, -64, True), (full[:65], 100, -72, False),
        (full[:65], 255, -72, True), (full[:72], 255, -72, True), (full[:73], 255, -80, True), (full[:80], 255, -80, True), (full[:81], 255, -88, True), (full[:88], 255, -88, True), (full[:89], 255, -100, True), (full[:99], 255, -100, True),
        (full[:34], 300, -36, True), (full[:99], 421, -100, True),
        (with_n[:13], 100, -13, False), (with_n[:13], 30, -13, False), (with_n[:16], 100, -16, False),      # an N in a short adapter is one more pattern of the pair table (round 6)
        (with_n[:24], 100, -24, True), (with_n[:34], 100, -36, True),                                          # ... and of a long one: the same instances as an adapter without N
        (with_n[:48], 150, -48, True), (with_n[:52], 150, -52, True), (with_n[:64], 150, -64, True), (with_n[:70], 200, -72, True), (with_n[:99], 300, -100, True),
        (b"ACGTRYKMSWBDHVXZACGT", 100, 32, False), (b"ACGTRY" * 6, 100, -36, True),                             # more than six distinct bytes in a long adapter: the general form; six: still the table
        (with_n[:13], 300, -13, False),
    ]
    for ad, stride, amax, two in cases:
        b = np.ascontiguousarray(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=(3, stride)))
        q = np.full((3, stride), 70, dtype=np.uint8)
        emu.run_pipeline(b, q, None, oracle_params(dict(stages=1, adapter=ad, clip_min_len=5, clip_flags=4)))
        assert emu.last_plan() == (amax, two), (ad, stride, emu.last_plan(), (amax, two))
    monkeypatch.setenv("FXG_CLIP_K_ONE_PASS", "1")
    emu.run_pipeline(b, q, None, oracle_params(dict(stages=1, adapter=full[:34], clip_min_len=5, clip_flags=4)))
    assert emu.last_plan()[0] > 0                       # 300-base reads, 34 columns, one pass: only the general form describes them
    monkeypatch.delenv("FXG_CLIP_K_ONE_PASS")
    monkeypatch.setenv("FXG_NO_PACKED_CLIP", "1")
    emu.run_pipeline(b, q, None, oracle_params(dict(stages=1, adapter=full[:13], clip_min_len=5, clip_flags=4)))
    assert emu.last_plan() == (16, False)
