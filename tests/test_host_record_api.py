"""CPU tier: the libfastx-compatible record API of the host layer (fastx_read_next_record / fastx_write_record)
against the reference's reader/writer, through the GPU-free demo tool host/bin/fastx_copy.

Reference side: `fxref fastx_trimmer -f 1` is the identity transform around the reference's own
fastx_read_next_record + fastx_write_record (skipped where oracle/_ref is absent; the Galaxy inputs are
then still checked for round-tripping).
"""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from helpers import GOLDEN
from oracle import fxoracle_py as fo

HOSTBIN = os.path.join(ROOT, "fastx_toolkit_amd", "host", "bin")
REF = fo.ref_binary()


@pytest.fixture(scope="module")
def copy_tool():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "fastx_toolkit_amd", "host"), "bin/fastx_copy"])
    return os.path.join(HOSTBIN, "fastx_copy")


def _run(cmd, data):
    p = subprocess.run(cmd, input=data, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    return p.returncode, p.stdout, p.stderr


def _ref_identity(data, extra=()):
    return _run([REF, "fastx_trimmer", "-f", "1"] + list(extra), data)


CASES = [
    b"@r1\nACGTN\n+\nIIIII\n@r2 x y\nAC\n+r2 x y\nI#\n",
    b"@r1\r\nACGTN\r\n+\r\nIIIII\r\n@r2\r\nAC\r\n+\r\nII\r\n",                       # CRLF in, LF out (R2)
    b"@r1\nACGT\n+\nIIII",                                                            # no final newline
    b"@r1\nACGT\n+anything here\n40 40 -5 3\n@r2\nAC\n+\n10 -0\n",                    # numeric qualities (R6)
    b"@r1\nACGT\n+\nIIII\n@r2\nACG\n+\n1 2 3\n@r3\nAC\n+\nII\n",                      # encodings mixed per record
    b">s1\nACGTNNACGT\n>s2-17\nAC\n",                                                 # FASTA incl. collapsed id
    b"@r1\nACGT\nXthird line junk\nIIII\n",                                           # line 3 is not validated (R5)
    b"@r\rjunk\nACGT\rjunk\n+\rjunk\nIIII\rjunk\n",                                   # chomp cuts at the first CR
    b"@r1\nACGT\n+\n1 2 3 4294967297\n",                                               # strtol's long lands in an int: 1
    b"@r1\nACGT\n+\n10-5 3+4\n@r2\nACG\n+\n\t1 \x0b2  +3\n",                           # strtol tokens: a sign starts a new value, any isspace() separates
]
BAD = [
    b"", b"ACGT\n", b"@r1\nACGX\n+\nIIII\n", b"@r1\nacgt\n+\nIIII\n", b"@r1\n\n+\n\n", b"@r1\nACGT\n+\nIIII\n\n",
    b"@r1\nACGT\n+\nIII\x07\n", b"@r1\nACGT\n+\n", b"@r1\nACGT\n", b"@r1\nACGT\n+\n1 2 3\n", b"@r1\nACGT\n+\n1 2 x 4\n",
    b"@r1\nACGT\n+\n1 2 3 99\n", b"@r1\nACGT\n+\n1 2 3 4 \n", b"@r1\nACGT\n+\n1 2 3 --4\n", b">s1\nACGT\nACGT\n", b">s1\nACGT\n@r\nAC\n", b"@ok\nAC\n+\nII\n@r1\nAC\n+\nI\n",
]


@pytest.mark.parametrize("data", CASES)
def test_round_trip_matches_reference(copy_tool, data):
    rc, out, err = _run([copy_tool], data)
    assert rc == 0, err
    if REF:
        rrc, rout, _ = _ref_identity(data)
        assert (rc, out) == (rrc, rout)


@pytest.mark.parametrize("data", BAD)
def test_malformed_input_fails_like_reference(copy_tool, data):
    rc, out, err = _run([copy_tool], data)
    assert rc == 1
    if REF:
        rrc, rout, rerr = _ref_identity(data)
        assert (rc, out) == (rrc, rout)                       # same exit code, same partial output
        assert err.split(b": ", 1)[-1] == rerr.split(b": ", 1)[-1]   # same message (program name differs)


def test_quality_offset_and_galaxy_inputs(copy_tool):
    for name, extra in (("fastq_quality_trimmer.fastq", ["-Q", "64"]), ("fastx_clipper1.fastq", ["-Q", "64"]),
                        ("fastx_trimmer2.fastq", []), ("fastx_rev_comp1.fasta", []), ("fastx_trimmer1.fasta", [])):
        data = open(os.path.join(GOLDEN, "galaxy", name), "rb").read()
        rc, out, err = _run([copy_tool] + extra, data)
        assert rc == 0, err
        if REF:
            assert out == _ref_identity(data, extra)[1], name
        else:
            assert out.replace(b"\n", b"") != b""


def test_fuzz_text_vs_reference(copy_tool):
    if not REF:
        pytest.skip("oracle/_ref not built")
    rng = np.random.default_rng(5)
    for trial in range(25):
        recs = []
        for i in range(int(rng.integers(1, 60))):
            L = int(rng.integers(1, 80))
            s = rng.choice(np.frombuffer(b"ACGTN", np.uint8), size=L).tobytes()
            if rng.random() < 0.5:
                q = rng.integers(33, 127, size=L, dtype=np.uint8).tobytes()
            else:
                q = b" ".join(str(int(v)).encode() for v in rng.integers(-15, 94, size=L))
                if L == len(q):
                    q = rng.integers(33, 127, size=L, dtype=np.uint8).tobytes()
            eol = b"\r\n" if rng.random() < 0.2 else b"\n"
            recs.append(b"@read %d" % i + eol + s + eol + b"+" + (b"x" if rng.random() < 0.3 else b"") + eol + q + eol)
        data = b"".join(recs)
        if rng.random() < 0.3:                                  # damage one byte somewhere
            k = int(rng.integers(0, len(data)))
            data = data[:k] + bytes([int(rng.integers(1, 255))]) + data[k + 1:]
        rc, out, err = _run([copy_tool, "-v"], data)
        rrc, rout, rerr = _ref_identity(data, ["-v"])
        assert (rc, out) == (rrc, rout), trial
        if rc == 0:
            assert err == rerr


def test_overlong_identifier_lines_are_rejected_not_overflowed(copy_tool):
    """Lines of 24 999+ characters: the reference's fgets() would split them silently; this layer refuses them -- for the
    identifier and '+' lines too, whose FASTX buffers hold MAX_SEQ_LINE_LENGTH bytes."""
    long_id = b"x" * 30000
    for data in (b"@" + long_id + b"\nACGT\n+\nIIII\n", b"@r\nACGT\n+" + long_id + b"\nIIII\n", b">" + long_id + b"\nACGT\n",
                 b"@ok\nAC\n+\nII\n@" + long_id + b"\nACGT\n+\nIIII\n"):
        rc, out, err = _run([copy_tool], data)
        assert rc == 1 and b"longer than" in err
    rc, out, err = _run([copy_tool], b"@" + b"x" * 24000 + b"\nACGT\n+" + b"y" * 24000 + b"\nIIII\n")
    assert rc == 0 and out == b"@" + b"x" * 24000 + b"\nACGT\n+" + b"y" * 24000 + b"\nIIII\n"
