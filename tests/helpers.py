"""Shared helpers for the parity tests (oracle side = checker only)."""
import hashlib
import os

import numpy as np

from oracle import fxoracle_py as fo

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KEYS = ("res", "out_bases", "out_qual", "out_len", "kept_index")


def md5(b):
    return hashlib.md5(b).hexdigest()


def oracle_params(d):
    return fo.make_params(**d)


def assert_same(o, e, what=""):
    """o = oracle result dict, e = engine/emulator result dict."""
    for k in KEYS:
        if o.get(k) is None:
            continue
        assert e.get(k) is not None, "%s: %s missing" % (what, k)
        assert o[k].shape == e[k].shape, "%s: %s shape %s vs %s" % (what, k, o[k].shape, e[k].shape)
        if not np.array_equal(o[k], e[k]):
            i = int(np.nonzero(o[k] != e[k])[0][0])
            raise AssertionError("%s: %s differs first at %d: oracle %r engine %r" % (what, k, i, o[k][i], e[k][i]))
    if e.get("out_off") is not None:
        ol = o["out_len"].astype(np.uint64)
        off = np.concatenate([[0], np.cumsum(ol)[:-1]]).astype(np.uint64) if len(ol) else np.zeros(0, np.uint64)
        assert np.array_equal(off, e["out_off"]), "%s: out_off" % what
    sel = list(range(13)) + [13, 14, 16]
    assert np.array_equal(o["counters"][sel], e["counters"][sel]), "%s: counters %s vs %s" % (what, o["counters"][sel], e["counters"][sel])


def text_through(run, text, params, qoffset=33):
    """FASTQ text -> SoA (oracle parser) -> run(bases, qual, lens, params) -> FASTQ text (oracle formatter)."""
    p = fo.parse_fastq(text, qoffset)
    r = run(p["bases"], p["qual"], p["lens"], params)
    return fo.format_fastq(text, p["names"], r["out_bases"], r["out_qual"], r["out_len"], r["kept_index"]), r


def random_batch(rng, n, stride, lmin, lmax, fixed=False, p_n=0.02, adapter=None):
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    b = rng.choice(acgt, size=(n, stride))
    b[rng.random((n, stride)) < p_n] = ord("N")
    q = rng.integers(33, 33 + 42, size=(n, stride), dtype=np.uint8)
    for i in range(n):
        d = int(rng.integers(0, stride + 1))
        q[i, d:] = rng.integers(33, 33 + 15, size=stride - d)
    lens = None if fixed else rng.integers(lmin, lmax + 1, size=n).astype(np.uint16)
    if adapter is not None:
        ad = np.frombuffer(adapter, dtype=np.uint8)
        for i in range(n):
            if rng.random() < 0.6:
                L = lmax if fixed else int(lens[i])
                pos = int(rng.integers(0, L + 1))
                k = min(len(ad), L - pos)
                a2 = ad.copy()
                if rng.random() < 0.3:
                    a2[int(rng.integers(0, len(ad)))] = rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8))
                b[i, pos:pos + k] = a2[:k]
    return np.ascontiguousarray(b), np.ascontiguousarray(q), lens


def fuzz_cases(seed, trials, clip_trials):
    """Yields (name, bases, qual, lens, fixed_len, params_dict) covering ragged lengths, tiny reads, odd strides, all flags."""
    rng = np.random.default_rng(seed)
    for trial in range(trials):
        stride = int(rng.choice([1, 2, 7, 15, 16, 17, 33, 36, 64, 100, 150, 151, 250, 300, 1000]))
        lmin = max(1, int(rng.integers(1, stride + 1)))
        fixed = rng.random() < 0.4
        n = int(rng.integers(1, 900))
        b, q, lens = random_batch(rng, n, stride, lmin, stride, fixed)
        fl = int(rng.integers(1, stride + 1)) if fixed else None
        t = int(rng.integers(-5, 45)) or 7
        ml = int(rng.integers(0, stride + 2))
        mq, pc = int(rng.integers(0, 45)), int(rng.integers(0, 101))
        qo = int(rng.choice([33, 33, 64, 30]))
        for st in (2, 4, 6):
            yield ("t%d.st%d.s%d.n%d" % (trial, st, stride, n), b, q, lens, fl,
                   dict(stages=st, qt_threshold=t, qt_min_len=ml, qf_min_quality=mq, qf_min_percent=pc, qoffset=qo))
        f, l = int(rng.integers(1, stride + 2)), int(rng.integers(0, stride + 3))
        for st in (8, 16, 24):
            yield ("t%d.st%d.f%d.l%d.s%d" % (trial, st, f, l, stride), b, q, lens, fl, dict(stages=st, ft_first=f, ft_last=l))
        for st in (32, 40):
            yield ("t%d.st%d.s%d" % (trial, st, stride), b, q, lens, fl,
                   dict(stages=st, ft_trim_end=int(rng.integers(1, stride + 2)), ft_min_len=int(rng.integers(0, stride + 1))))
        yield ("t%d.mask.s%d" % (trial, stride), b, q, lens, fl,
               dict(stages=64, mask_min_quality=int(rng.integers(-5, 45)), mask_char=str(rng.choice(list("N.x"))), qoffset=qo))
        b2 = b.copy()
        for i in range(0, n, 5):                                   # plant artifact-like reads (one base everywhere but <= 4 places)
            b2[i, :] = ord("ACGT"[i % 4])
            b2[i, :min(stride, int(rng.integers(0, 6)))] = ord("ACGTN"[int(rng.integers(0, 5))])
        yield ("t%d.artifacts.s%d" % (trial, stride), b2, q, lens, fl, dict(stages=128))
        yield ("t%d.nfilter.s%d" % (trial, stride), b, q, lens, fl, dict(stages=256, nf_keep_n=trial % 3 == 0))
    adapters = [b"AGATCGGAAGAGC", b"CCTTAAGG", b"CAATTGGTTAATCCCCCTATATA", b"ACGT", b"TGGAATTCTCGGGTGCCAAGGAACTCCAGTCAC",
                b"ANNTCGNA", b"A" * 40 + b"CGT" * 8, b"ACGTTGCA" * 9]
    for trial in range(clip_trials):
        ad = adapters[trial % len(adapters)]
        stride = int(rng.choice([5, 13, 20, 36, 50, 75, 100, 151]))
        n = int(rng.integers(1, 700))
        b, q, _ = random_batch(rng, n, stride, stride, stride, True, adapter=ad)
        flags = int(rng.integers(0, 16))
        d = int(rng.integers(0, 3)) * int(rng.integers(0, 8))
        kd = d + len(ad) if d > 0 else 0
        yield ("clip%d.a%d.s%d.fl%d" % (trial, len(ad), stride, flags), b, q, None, stride,
               dict(stages=1, adapter=ad, clip_min_len=int(rng.integers(0, 25)), clip_keep_delta=kd,
                    clip_min_adapter_len=int(rng.choice([0, 0, 3, 8])), clip_flags=flags))
        yield ("clip7_%d" % trial, b, q, None, stride,
               dict(stages=7, adapter=ad, clip_min_len=5, clip_flags=flags & 7, qt_threshold=15, qt_min_len=4,
                    qf_min_quality=12, qf_min_percent=60))


# Command lines the reference rejects (exit 1 + a message), with an input that is otherwise fine: shared by the CPU tier (emulation
# stub) and the GPU tier, both of which compare exit code, stdout and message with the real libfastx driver (oracle/_ref/fxref).
_REC = b"@r\nA\n+\nI\n"
FLAG_ERROR_CASES = [
    (["fastq_quality_trimmer"], _REC),                                   # -t missing (fastq_quality_trimmer.c:82-83)
    (["fastq_quality_trimmer", "-t", "0"], _REC),
    (["fastq_quality_trimmer", "-t", "20", "-l", "-1"], _REC),           # :60-62 -- strtoul("-1") lands negative in an int
    (["fastq_quality_trimmer", "-t", "20", "-l", "-300"], _REC),
    (["fastq_quality_filter", "-p", "0"], b""),
    (["fastq_quality_filter", "-p", "101", "-q", "5"], _REC),
    (["fastq_quality_filter", "-p", "-5", "-q", "5"], _REC),
    (["fastx_trimmer", "-f", "2", "-t", "3"], _REC),
    (["fastx_trimmer", "-f", "0"], _REC),
    (["fastx_trimmer", "-f", "-2"], _REC),
    (["fastx_trimmer", "-l", "25000"], _REC),
    (["fastx_trimmer", "-l", "0"], _REC),
    (["fastx_trimmer", "-t", "0"], _REC),
    (["fastx_trimmer", "-t", "3", "-m", "0"], _REC),
    (["fastx_trimmer", "-t", "25000"], _REC),
    (["fastx_clipper", "-M", "0"], _REC),
    (["fastx_clipper", "-M", "-4"], _REC),
    (["fastx_clipper", "-d", "-3"], _REC),                                # fastx_clipper.cpp:121-123
    (["fastq_masker", "-r", "xy"], _REC),
    (["fastq_masker", "-q", "-50"], _REC),                                # fastq_masker.c:62-63
    (["fastq_masker", "-r", ""], _REC),
    (["fastq_quality_trimmer", "-t", "20"], b""),                        # empty input (R1)
    (["fastq_quality_trimmer", "-t", "20"], b">fa\nAC\n"),              # FASTQ only
    (["fastq_quality_trimmer", "-t", "20", "-Q", "64"], b"@r\nA\n+\n!\n"),
    (["fastq_quality_trimmer", "-t", "20", "-x"], _REC),                 # unknown option: getopt's message, then the usage hint
    (["fastx_artifacts_filter", "-q", "3"], _REC),
]


def adversarial_clip_cases(long_adapters):
    """Yields (name, bases, qual, params_dict): inputs built to stress the clip kernels' bound on the best path's length -- every adapter
    length 1..16 (all packed buckets of the two-pass form) or, with long_adapters, 17..99 (every bucket of the one-pass in-place form);
    low-complexity reads and adapters (long runs of ties), N-rich reads and adapters, adapters repeated along the read, reads shorter
    than the adapter, best cells in the first / last rows, an insertion / deletion right before a planted adapter."""
    rng = np.random.default_rng(11)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    for trial in range(128 if long_adapters else 96):
        alen = 1 + trial % 16
        if long_adapters:                                                       # both ends of every bucket (44, 52, 60, 72, 88: round 6)
            alen = [17, 20, 21, 24, 25, 28, 29, 32, 33, 36, 37, 40, 41, 44, 45, 48, 49, 52, 53, 56, 57, 60, 61, 64, 65, 72, 73, 80, 81, 88, 89, 99][trial % 32]
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
            ad = (b"AGATCGGAAGAGCACACGTCTGAACTCCAGTCACATCACGATCTCGTATGCCGTCTTCTGCTTGAAAAAAAAAAGGGGGGGGGGCCCCCCCCCCTTTTTTTTTT")[:alen]
        else:
            ad = bytes(rng.choice(acgt[:2], size=alen))
        if ad.count(b"N") == alen:
            ad = b"A" + ad[1:]
        stride = int(rng.choice([3, 8, 17, 30, 64, 100, 150, 255, 300, 421]))    # beyond 255: only the forms with a relative path start
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
            yield "clip2.t%d.a%s.s%d.f%d" % (trial, ad.decode(), stride, flags), b, q, pd


def first_n_cases(seed=41):
    """Inputs for the clipper's -n rule (a read with an N before its clip point is dropped unless -n): the first N in every position class of the
    dword scan -- byte 0..3 of a dword, the last valid base, the byte just PAST the end of a ragged read (must not count), a 0x4F / 0x4D byte right
    above an N (what the zero-byte trick could mistake), no N at all -- for strides that are and are not multiples of 4, fixed and ragged,
    short adapters (register form) and a 34-base one (checkpoint form).  Yields (name, bases, quals, lens, fixed_len, params)."""
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    for stride in (40, 41, 42, 43, 100, 150, 151):
        for ragged in (False, True):
            for ad in (b"AGATCGGAAGAGC", b"AGATCGGAAGAGCACACGTCTGAACTCCAGTCAC"):
                n = 192
                b = rng.choice(acgt, size=(n, stride)).astype(np.uint8)
                lens = rng.integers(max(1, stride - 9), stride + 1, size=n).astype(np.uint16) if ragged else None
                for i in range(n):
                    L = int(lens[i]) if ragged else stride
                    kind = i % 8
                    if kind == 0: continue                                   # no N
                    if kind == 1: pos = [int(rng.integers(0, L))]
                    elif kind == 2: pos = [L - 1]
                    elif kind == 3: pos = [L] if L < stride else [L - 1]     # just past the end (ragged) -- not an N of the read
                    elif kind == 4: pos = sorted(int(x) for x in rng.integers(0, L, size=3))
                    elif kind == 5: pos = [4 * int(rng.integers(0, max(1, L // 4))) + int(rng.integers(0, 4))]
                    elif kind == 6: pos = [int(rng.integers(0, L))]
                    else: pos = [0]
                    for x in pos:
                        if x < stride: b[i, x] = ord("N")
                    if kind == 6 and pos[0] + 1 < stride: b[i, pos[0] + 1] = ord("O") if i % 16 < 8 else ord("M")   # 'N' ^ 'O' = 1: the borrow case
                q = rng.integers(33, 74, size=(n, stride), dtype=np.uint8)
                pd = dict(stages=1, adapter=ad, clip_min_len=5, clip_flags=0)
                yield ("first_n.s%d.%s.a%d" % (stride, "ragged" if ragged else "fixed", len(ad)), np.ascontiguousarray(b), q, lens, None if ragged else stride, pd)



def odd_alphabet_clip_cases(seed=77):
    """Clipper batches whose bytes are not all ACGTN (lower case, IUPAC codes, arbitrary letters, in the adapter and in the reads): the reference compares
    bytes with `==` and treats only 'N' as neutral (sequence_alignment.h:147-169), whatever the alphabet -- the pair table of the register two-pass
    instances (fxg_kernels.h: fxg_clip_ptab_build) must serve every byte value the same way.  Yields (name, bases, qual, fixed_len, params_dict)."""
    rng = np.random.default_rng(seed)
    adapters = [b"agatcggaagagc", b"AGRYCGGAWGAGC", b"ACGTRYKMSWBDHVXZ", b"AcGtAcGtAcG", b"XXXXXXXXXXXXX", b"ACGTacgtACGTa", b"TTTTTTTTTTTTTTTT", b"A", b"zQ",
                b"agatcggaagagcacacgtctgaactccagtcac", b"AGATCGGAAGAGCACACGTCTGAACTCCAGTCACNNNNNNATCTCGTATGCCGTCTTCTGCTTG", b"AGRYCGGAWGAGCACACGTCTGAAC", b"AcGtNNAcGtAcGtAcGtAcGt" * 2]
    # (the long ones: four distinct bytes in lower case -> the pair table of the 36-column instance; N columns in a 64-column one; seven distinct bytes -> the general form;
    #  six distinct bytes with N columns -> still the table)
    for k, ad in enumerate(adapters):
        for stride in (13, 36, 100, 151):
            n = int(rng.integers(50, 500))
            b, q, _ = random_batch(rng, n, stride, stride, stride, True, adapter=ad)
            alpha = np.frombuffer(bytes(sorted(set(ad))) + b"acgtnXRY*", dtype=np.uint8)
            hit = rng.random((n, stride)) < 0.15
            b[hit] = rng.choice(alpha, size=int(hit.sum()))
            for i in range(0, n, 7):                         # a clean copy of the adapter somewhere in every seventh read
                pos = int(rng.integers(0, stride))
                m = min(len(ad), stride - pos)
                b[i, pos:pos + m] = np.frombuffer(ad, dtype=np.uint8)[:m]
            flags = int(rng.integers(0, 16))
            yield ("odd%d.s%d.fl%d" % (k, stride, flags), np.ascontiguousarray(b), q, stride,
                   dict(stages=1, adapter=ad, clip_min_len=int(rng.integers(0, 12)), clip_flags=flags, clip_min_adapter_len=int(rng.choice([0, 0, 4]))))
