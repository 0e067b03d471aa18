"""CPU tier: the oracle (plain-C restatement) against the reference's own known answers.

1. every Galaxy known-answer pair of the five hot tools (data files the reference ships);
2. md5 sums / counts of reference output on seeded synthetic inputs (tests/golden/cases.json, produced by
   tests/golden/make_golden.py from the real reference libfastx; the cfg1..cfg5 sums equal SURVEY.md 8d).
"""
import os

import numpy as np
import pytest

from helpers import GOLDEN, md5, oracle_params, text_through
from oracle import fxoracle_py as fo


def _read(name):
    return open(os.path.join(GOLDEN, "galaxy", name), "rb").read()


def _fasta_or_numeric(text):
    """Tiny reader for the FASTA / numeric-quality Galaxy inputs (qualities encoded as value+33 bytes)."""
    lines = text.split(b"\n")
    recs = []
    if text[:1] == b">":
        for i in range(0, len(lines) - 1, 2):
            if lines[i]:
                recs.append((lines[i], lines[i + 1], None, None))
    else:
        for i in range(0, len(lines) - 3, 4):
            if lines[i]:
                recs.append((lines[i], lines[i + 1], lines[i + 2], [int(x) for x in lines[i + 3].split()]))
    return recs


def _run_records(recs, params):
    n = len(recs)
    stride = max(len(r[1]) for r in recs)
    b = np.zeros((n, stride), np.uint8)
    q = np.zeros((n, stride), np.uint8) if recs[0][3] is not None else None
    lens = np.zeros(n, np.uint16)
    for i, (_, s, _, ql) in enumerate(recs):
        b[i, :len(s)] = np.frombuffer(s, np.uint8)
        lens[i] = len(s)
        if ql is not None:
            assert len(ql) == len(s)
            q[i, :len(s)] = np.array(ql) + 33
    r = fo.run_pipeline(b, q, lens, params)
    out, pos = [], 0
    for k, idx in enumerate(r["kept_index"]):
        l = int(r["out_len"][k])
        name, _, name2, ql = recs[idx]
        out.append(name)
        out.append(r["out_bases"][pos:pos + l].tobytes())
        if ql is not None:
            out.append(b"+" + name2[1:])
            out.append(b" ".join(str(int(v) - 33).encode() for v in r["out_qual"][pos:pos + l]))
        pos += l
    return b"\n".join(out) + b"\n"


def test_galaxy_known_answers(cases):
    for g in cases["galaxy"]:
        inp, exp = _read(g["input"]), _read(g["expect"])
        p = oracle_params(g["params"])
        ascii_fastq = inp[:1] == b"@" and len(inp.split(b"\n")[3]) == len(inp.split(b"\n")[1])
        if ascii_fastq and g.get("fasta_out"):          # fastq_to_fasta: FASTA writer, optional renaming to the output index
            pr = fo.parse_fastq(inp, g["params"].get("qoffset", 33))
            r = fo.run_pipeline(pr["bases"], pr["qual"], pr["lens"], p)
            no, nl = pr["names"][0], pr["names"][1]
            out, pos = [], 0
            for k, idx in enumerate(r["kept_index"]):
                l = int(r["out_len"][k])
                name = str(k + 1).encode() if g.get("rename") else inp[int(no[idx]):int(no[idx]) + int(nl[idx])]
                out.append(b">" + name + b"\n" + r["out_bases"][pos:pos + l].tobytes() + b"\n")
                pos += l
            got = b"".join(out)
        elif ascii_fastq:
            got, _ = text_through(fo.run_pipeline, inp, p, qoffset=g["params"].get("qoffset", 33))
        else:
            got = _run_records(_fasta_or_numeric(inp), p)
        assert got == exp, g["name"]


@pytest.mark.parametrize("pick", ["small", "cfg"])
def test_synthetic_reference_md5(cases, pick):
    for c in cases["synthetic"]:
        big = c["n"] > 100000
        if (pick == "small") == big or c["n"] > 200000:
            continue
        text = fo.synth_fastq(c["seed"], 0, c["n"], c["L"], c["adapter"])
        assert md5(text) == c["input_md5"], c["name"]
        got, r = text_through(fo.run_pipeline, text, oracle_params(c["params"]))
        assert (int(r["counters"][fo.C_KEPT]), int(r["counters"][fo.C_KEPT_BASES])) == (c["kept"], c["kept_bases"]), c["name"]
        assert md5(got) == c["output_md5"], c["name"]


def test_variable_length_reference_outputs(cases):
    """Inputs are reference-trimmed (ragged) reads; the clipper case exercises the history quirk N3."""
    for c in cases["varlen"]:
        text = open(os.path.join(GOLDEN, "synthetic", c["name"] + ".fq"), "rb").read()
        exp = open(os.path.join(GOLDEN, "synthetic", c["name"] + ".out"), "rb").read()
        assert md5(text) == c["input_md5"]
        got, _ = text_through(fo.run_pipeline, text, oracle_params(c["params"]))
        assert got == exp, c["name"]


def test_reader_rules():
    ok = b"@r1\nACGTN\n+\nIIIII\n@r2 x\r\nAC\r\n+r2 x\r\nII\r\n@r3\nA\n+\nI"    # CRLF accepted, missing final newline fine (R2)
    p = fo.parse_fastq(ok)
    assert p["n"] == 3 and list(p["lens"]) == [5, 2, 1]
    for bad in (b"", b">x\nAC\n", b"@r\nACGX\n+\nIIII\n", b"@r\nacgt\n+\nIIII\n", b"@r\n\n+\n\n",
                b"@r\nAC\n+\nII\n\n", b"@r\nAC\n+\nI\x05\n", b"@r\nAC\n+\n"):
        with pytest.raises(ValueError):
            fo.parse_fastq(bad)


def test_quality_stats_galaxy_known_answer():
    """fastx_quality_stats restatement vs galaxy/test-data/fastq_stats1.{fastq,out} (Illumina-1.3 qualities: -Q 64)."""
    from helpers import GOLDEN
    import os
    text = open(os.path.join(GOLDEN, "galaxy", "fastq_stats1.fastq"), "rb").read()
    p = fo.parse_fastq(text, qoffset=64)
    qs = fo.QStats()
    qs.add(p["bases"], p["qual"], p["lens"], qoffset=64)
    assert qs.text(False) == open(os.path.join(GOLDEN, "galaxy", "fastq_stats1.out"), "rb").read()
    assert qs.text(True).startswith(b"cycle\tmax_count\tALL_count")
    qs.close()
