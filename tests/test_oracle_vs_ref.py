"""CPU tier: the C restatement against the REAL reference libfastx (oracle/_ref/fxref) on fuzzed FASTQ text.

Skipped where oracle/_ref was never built (it needs /root/reference); the md5 fixtures in
test_oracle_golden.py carry the same evidence to machines without it.
"""
import subprocess

import numpy as np
import pytest

from helpers import oracle_params, text_through
from oracle import fxoracle_py as fo

REF = fo.ref_binary()
pytestmark = pytest.mark.skipif(REF is None, reason="oracle/_ref/fxref not built")


def _ref(text, chain):
    for cmd in chain:
        p = subprocess.run([REF] + cmd, input=text, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert p.returncode == 0, p.stderr
        text = p.stdout
    return text


def _fastq(rng, n, lmin, lmax, adapter=None):
    out = []
    for i in range(n):
        L = int(rng.integers(lmin, lmax + 1))
        s = rng.choice(np.frombuffer(b"ACGTN", np.uint8), size=L, p=[.24, .24, .24, .24, .04])
        if adapter is not None and rng.random() < 0.6:
            pos = int(rng.integers(0, L + 1))
            k = min(len(adapter), L - pos)
            s[pos:pos + k] = np.frombuffer(adapter, np.uint8)[:k]
        q = rng.integers(33, 75, size=L, dtype=np.uint8)
        q[int(rng.integers(0, L + 1)):] = 35
        out.append(b"@r%d\n%s\n+\n%s\n" % (i, s.tobytes(), q.tobytes()))
    return b"".join(out)


def test_quality_tools_vs_reference():
    rng = np.random.default_rng(1)
    for trial in range(12):
        text = _fastq(rng, 300, 1, 120)
        t, l = int(rng.integers(-3, 40)) or 5, int(rng.integers(0, 60))
        q, p = int(rng.integers(0, 42)), int(rng.integers(1, 101))
        exp = _ref(text, [["fastq_quality_trimmer", "-t", str(t), "-l", str(l)]])
        got, _ = text_through(fo.run_pipeline, text, oracle_params(dict(stages=2, qt_threshold=t, qt_min_len=l)))
        assert got == exp
        exp = _ref(text, [["fastq_quality_filter", "-q", str(q), "-p", str(p)]])
        got, _ = text_through(fo.run_pipeline, text, oracle_params(dict(stages=4, qf_min_quality=q, qf_min_percent=p)))
        assert got == exp
        exp = _ref(text, [["fastq_quality_trimmer", "-t", str(t), "-l", str(max(l, 1))], ["fastq_quality_filter", "-q", str(q), "-p", str(p)]])
        got, _ = text_through(fo.run_pipeline, text, oracle_params(dict(stages=6, qt_threshold=t, qt_min_len=max(l, 1), qf_min_quality=q, qf_min_percent=p)))
        assert got == exp


def test_trimmer_revcomp_vs_reference():
    rng = np.random.default_rng(2)
    for trial in range(12):
        text = _fastq(rng, 300, 1, 90)
        f, l = int(rng.integers(1, 60)), int(rng.integers(1, 100))
        exp = _ref(text, [["fastx_trimmer", "-f", str(f), "-l", str(l)]])
        got, _ = text_through(fo.run_pipeline, text, oracle_params(dict(stages=16, ft_first=f, ft_last=l)))
        assert got == exp
        exp = _ref(text, [["fastx_reverse_complement"], ["fastx_trimmer", "-f", str(f), "-l", str(l)]])
        got, _ = text_through(fo.run_pipeline, text, oracle_params(dict(stages=24, ft_first=f, ft_last=l)))
        assert got == exp
        tn, m = int(rng.integers(1, 50)), int(rng.integers(1, 40))
        exp = _ref(text, [["fastx_trimmer", "-t", str(tn), "-m", str(m)]])
        got, _ = text_through(fo.run_pipeline, text, oracle_params(dict(stages=32, ft_trim_end=tn, ft_min_len=m)))
        assert got == exp


def test_masker_and_artifacts_vs_reference():
    rng = np.random.default_rng(6)
    for trial in range(10):
        text = _fastq(rng, 300, 1, 90)
        lines = text.split(b"\n")
        for k in range(1, len(lines) - 1, 28):                      # plant homopolymer-ish reads
            L = len(lines[k])
            lines[k] = (b"A" * L)[:L - min(L, trial % 5)] + lines[k][L - min(L, trial % 5):]
        text = b"\n".join(lines)
        mq, ch = int(rng.integers(-3, 45)), str(rng.choice(list("N.x")))
        exp = _ref(text, [["fastq_masker", "-q", str(mq), "-r", ch]])
        got, _ = text_through(fo.run_pipeline, text, oracle_params(dict(stages=64, mask_min_quality=mq, mask_char=ch)))
        assert got == exp
        exp = _ref(text, [["fastx_artifacts_filter"]])
        got, r = text_through(fo.run_pipeline, text, oracle_params(dict(stages=128)))
        assert got == exp
    assert int(r["counters"][16]) > 0


def test_clipper_vs_reference_fixed_and_ragged():
    rng = np.random.default_rng(3)
    ads = [b"AGATCGGAAGAGC", b"CCTTAAGG", b"CAATTGGTTAATCCCCCTATATA", b"ACGT", b"ANNTCGNA"]
    for trial in range(15):
        ad = ads[trial % len(ads)]
        ragged = trial % 3 == 0
        text = _fastq(rng, 250, 8 if ragged else 50, 70 if ragged else 50, adapter=ad)
        flags, argv = 0, ["fastx_clipper", "-a", ad.decode()]
        for bit, sw in ((1, "-c"), (2, "-C"), (4, "-n"), (8, "-k")):
            if rng.random() < 0.4:
                flags |= bit
                argv.append(sw)
        ml = int(rng.integers(0, 20))
        argv += ["-l", str(ml)]
        exp = _ref(text, [argv])
        got, _ = text_through(fo.run_pipeline, text, oracle_params(dict(stages=1, adapter=ad, clip_min_len=ml, clip_flags=flags)))
        assert got == exp, (trial, argv)


def test_aligner_fields_vs_reference():
    rng = np.random.default_rng(4)
    for _ in range(40):
        q = rng.choice(np.frombuffer(b"ACGTN", np.uint8), size=int(rng.integers(1, 60))).tobytes()
        t = rng.choice(np.frombuffer(b"ACGTN", np.uint8), size=int(rng.integers(1, 30))).tobytes()
        out = subprocess.run([REF, "align", q.decode(), t.decode()], stdout=subprocess.PIPE).stdout.split()
        r = fo.align(q, t)
        assert [int(x) for x in out] == [r.query_start, r.query_end, r.target_start, r.target_end, r.matches, r.mismatches,
                                         r.neutral_matches, r.gaps]


def test_quality_stats_vs_reference():
    """fastx_quality_stats: the restatement's two report formats against the real reader + the driver's restated tool body."""
    rng = np.random.default_rng(4)
    for trial in range(8):
        text = _fastq(rng, int(rng.integers(1, 400)), 1, int(rng.integers(1, 130)))
        p = fo.parse_fastq(text)
        qs = fo.QStats()
        qs.add(p["bases"], p["qual"], p["lens"], qoffset=33)
        for new in (False, True):
            exp = _ref(text, [["fastx_quality_stats"] + (["-N"] if new else [])])
            assert qs.text(new) == exp, (trial, new)
        qs.close()
