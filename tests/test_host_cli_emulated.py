"""CPU tier: the complete host layer of the command-line tools (flag parsing, block reader, worker threads, batch packing,
formatting, reports, error handling) against the reference driver, WITHOUT a GPU.

The tools are run with tests/emu/stub/libfxg.so first on LD_LIBRARY_PATH: a test-only stand-in for the engine library that
executes the kernels' per-thread code through the serial CPU emulator.  The product library has no CPU path; this stub is
never installed next to it.  The same comparisons run against the real GPU engine in tests/test_gpu_cli.py.
"""
import os
import subprocess
import sys
import time

import numpy as np
import pytest

from conftest import ROOT
from helpers import GOLDEN, md5
from oracle import fxoracle_py as fo

HOST = os.path.join(ROOT, "fastx_toolkit_amd", "host")
STUB_DIR = os.path.join(ROOT, "tests", "emu", "stub")
REF = fo.ref_binary()


@pytest.fixture(scope="module")
def tools():
    import emu_py
    assert os.path.samefile(emu_py.build_stub(), STUB_DIR)
    from fastx_toolkit_amd import build as b
    b.build_engine()          # the tools link against the real library's soname; the stub replaces it at run time only
    subprocess.check_call(["make", "-s", "-C", HOST])
    return os.path.join(HOST, "bin")


def _run(cmd, data, threads="4", buf_mb=None, extra_env=None):
    env = dict(os.environ, LD_LIBRARY_PATH=STUB_DIR, FXH_THREADS=threads)
    if buf_mb:
        env["FXH_READ_BUFFER_MB"] = buf_mb
    env.update(extra_env or {})
    p = subprocess.run(cmd, input=data, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=120)
    return p.returncode, p.stdout, p.stderr


def _msg(err):
    return err.split(b": ", 1)[-1]


def test_galaxy_known_answers_through_the_host_layer(tools, cases):
    for g in cases["galaxy"]:
        inp = open(os.path.join(GOLDEN, "galaxy", g["input"]), "rb").read()
        exp = open(os.path.join(GOLDEN, "galaxy", g["expect"]), "rb").read()
        rc, out, err = _run([os.path.join(tools, g["cmd"][0])] + g["cmd"][1:], inp)
        assert rc == 0, err
        assert out == exp, g["name"]


def test_small_synthetic_cases_md5(tools, cases):
    for c in cases["synthetic"]:
        if c["n"] > 3000:
            continue
        text = fo.synth_fastq(c["seed"], 0, c["n"], c["L"], c["adapter"])
        for cmd in c["chain"]:
            rc, text, err = _run([os.path.join(tools, cmd[0])] + cmd[1:], text)
            assert rc == 0, err
        assert md5(text) == c["output_md5"], c["name"]


@pytest.mark.skipif(REF is None, reason="oracle/_ref/fxref not built")
def test_fuzz_every_tool_vs_reference(tools):
    rng = np.random.default_rng(21)
    ad = b"AGATCGGAAGAGC"
    for trial in range(12):
        L = Lmax = int(rng.integers(20, 70))
        recs = []
        for i in range(int(rng.integers(30, 250))):
            if trial % 3 == 1:
                L = int(rng.integers(8, Lmax + 1))   # ragged input: the clipper then depends on the reads before (N3)
            s = rng.choice(np.frombuffer(b"ACGTN", np.uint8), size=L, p=[.24, .24, .24, .24, .04])
            if rng.random() < 0.5:
                pos = int(rng.integers(0, L + 1)); k = min(len(ad), L - pos)
                s[pos:pos + k] = np.frombuffer(ad, np.uint8)[:k]
            if i % 9 == 0:
                s[:] = ord("ACGT"[i % 4]); s[:int(rng.integers(0, 5))] = ord("C")
            q = rng.integers(33, 75, size=L, dtype=np.uint8); q[int(rng.integers(0, L + 1)):] = 36
            recs.append(b"@r%d some text\n%s\n+\n%s\n" % (i, s.tobytes(), q.tobytes()))
        data = b"".join(recs)
        if trial % 4 == 3:      # malformed record in the middle: earlier output must still appear, exit code 1
            k = len(b"".join(recs[:len(recs) // 2])) + 12
            data = data[:k] + b"!" + data[k + 1:]
        argvs = [["fastq_quality_trimmer", "-t", str(int(rng.integers(5, 40))), "-l", str(int(rng.integers(0, 50))), "-v"],
                 ["fastq_quality_filter", "-q", str(int(rng.integers(5, 40))), "-p", str(int(rng.integers(1, 101))), "-v"],
                 ["fastx_trimmer", "-f", str(int(rng.integers(1, 30))), "-l", str(int(rng.integers(30, 100))), "-v"],
                 ["fastx_trimmer", "-t", str(int(rng.integers(1, 30))), "-m", str(int(rng.integers(1, 60))), "-v"],
                 ["fastx_reverse_complement", "-v"],
                 ["fastq_masker", "-q", str(int(rng.integers(0, 45))), "-r", str(rng.choice(list("N.x"))), "-v"],
                 ["fastx_artifacts_filter", "-v"],
                 ["fastq_to_fasta", "-v"] + (["-r"] if trial % 2 else []) + (["-n"] if trial % 3 == 0 else []),
                 ["fastx_clipper", "-a", ad.decode(), "-l", str(int(rng.integers(0, 25))), "-v"] + list(rng.choice(["-n", "-c", "-C", "-k"], size=2, replace=False)),
                 ["fastx_quality_stats"] + (["-N"] if trial % 2 else [])]
        for argv in argvs:
            rc, out, err = _run([os.path.join(tools, argv[0])] + argv[1:], data, threads=str([4, 1, 3][trial % 3]), buf_mb="1" if trial % 2 else None)
            rrc, rout, rerr = _run([REF] + argv, data)
            assert (rc, out) == (rrc, rout), (trial, argv)
            assert _msg(err) == _msg(rerr), (trial, argv)


def test_flag_errors_and_usage(tools):
    from helpers import FLAG_ERROR_CASES
    for argv, data in FLAG_ERROR_CASES:                       # exit code, stdout and message of the real libfastx driver
        rc, out, err = _run([os.path.join(tools, argv[0])] + argv[1:], data)
        rrc, rout, rerr = _run([REF] + argv, data)
        assert rc == 1 and (rc, out) == (rrc, rout), argv
        assert _msg(err) == _msg(rerr), (argv, err, rerr)
    if REF:                                                     # -D is the reference's dump again (round 5, host/fxh_clip_debug.c): see test_clipper_debug_dump
        assert _run([os.path.join(tools, "fastx_clipper"), "-a", "ACGT", "-D"], b"@r\nA\n+\nI\n")[:2] == _run([REF, "fastx_clipper", "-a", "ACGT", "-D"], b"@r\nA\n+\nI\n")[:2]
    assert _run([os.path.join(tools, "fastq_quality_trimmer")], b"@r\nA\n+\nI\n")[0] == 1          # missing -t
    assert _run([os.path.join(tools, "fastq_quality_filter"), "-p", "0"], b"")[0] == 1
    assert _run([os.path.join(tools, "fastx_trimmer"), "-f", "2", "-t", "3"], b"@r\nA\n+\nI\n")[0] == 1
    rc, out, _ = _run([os.path.join(tools, "fastx_clipper"), "-h"], b"")
    assert rc == 1 and out.startswith(b"usage: fastx_clipper")                                     # -h exits 1 (F5)
    assert _run([os.path.join(tools, "fastq_quality_trimmer"), "-t", "20"], b"")[0] == 1          # empty input is an error (R1)
    assert _run([os.path.join(tools, "fastq_quality_trimmer"), "-t", "20"], b">fa\nAC\n")[0] == 1  # FASTQ only


def test_quality_stats_golden_fasta_and_offsets(tools):
    """fastx_quality_stats: Galaxy known answer (-Q 64), FASTA input (record API, collapsed-read weights), -o file, both formats."""
    from helpers import GOLDEN
    inp = open(os.path.join(GOLDEN, "galaxy", "fastq_stats1.fastq"), "rb").read()
    exp = open(os.path.join(GOLDEN, "galaxy", "fastq_stats1.out"), "rb").read()
    rc, out, err = _run([os.path.join(tools, "fastx_quality_stats"), "-Q", "64"], inp)
    assert (rc, out) == (0, exp), err
    fasta = b">1-5\nACGTNACGT\n>2-3\nTTGCA\n>x\nGGGGGGGGGGGG\n"
    for extra in ([], ["-N"]):
        rc, out, err = _run([os.path.join(tools, "fastx_quality_stats")] + extra, fasta)
        rrc, rout, rerr = _run([REF, "fastx_quality_stats"] + extra, fasta)
        assert (rc, out) == (rrc, rout), (extra, err, rerr)
        rc, out, err = _run([os.path.join(tools, "fastx_quality_stats"), "-Q", "64"] + extra, inp, threads="3", buf_mb="1")
        rrc, rout, rerr = _run([REF, "fastx_quality_stats", "-Q", "64"] + extra, inp)
        assert (rc, out) == (rrc, rout), (extra, err, rerr)


def test_gzip_output_is_parallel_members(tools, tmp_path):
    """-z: the tools deflate their output themselves, one gzip member per MiB on worker threads; zcat must give the plain output."""
    import gzip
    text = fo.synth_fastq(31, 0, 40000, 100, False)                     # ~9 MB of FASTQ: several members, several threads
    plain = _run([os.path.join(tools, "fastq_quality_trimmer"), "-t", "20", "-l", "30"], text)
    for threads in ("1", "5"):
        rc, out, err = _run([os.path.join(tools, "fastq_quality_trimmer"), "-t", "20", "-l", "30", "-z"], text, threads=threads)
        assert rc == 0, err
        assert out[:2] == b"\x1f\x8b" and out.count(b"\x1f\x8b\x08") >= 3
        assert gzip.decompress(out) == plain[1]
    p = subprocess.run(["zcat"], input=out, stdout=subprocess.PIPE, timeout=60)
    assert p.returncode == 0 and p.stdout == plain[1]
    outp = tmp_path / "o.fq.gz"
    rc, out, err = _run([os.path.join(tools, "fastq_quality_filter"), "-q", "93", "-p", "100", "-z", "-o", str(outp)], text)   # nothing passes
    assert rc == 0 and gzip.decompress(outp.read_bytes()) == b""       # still a valid, empty gzip file
    head = b"\n".join(text.split(b"\n")[:4000]) + b"\n"                                                                       # 1000 whole records
    rc, out, err = _run([os.path.join(tools, "fastx_copy"), "-z"], head)                                                       # record API path
    assert rc == 0 and gzip.decompress(out) == _run([os.path.join(tools, "fastx_copy")], head)[1] == head


def _odd_inputs(rng):
    """Reader rules R1-R9 through the batch path: FASTA, numeric qualities, CRLF, no final newline, collapsed ids."""
    L = int(rng.integers(12, 60))
    recs_fq, recs_num, recs_fa = [], [], []
    for i in range(int(rng.integers(20, 300))):
        n = L if rng.random() < 0.6 else int(rng.integers(6, L + 1))
        s = rng.choice(np.frombuffer(b"ACGTN", np.uint8), size=n, p=[.24, .24, .24, .24, .04]).tobytes()
        q = rng.integers(0, 41, size=n)
        recs_fq.append(b"@r%d x\n%s\n+r%d\n%s\n" % (i, s, i, bytes((q + 33).astype(np.uint8))))
        recs_num.append(b"@r%d\n%s\n+\n%s\n" % (i, s, b" ".join(b"%d" % (int(v) - 5) for v in q)))
        recs_fa.append(b">%d-%d\n%s\n" % (i, int(rng.integers(1, 9)), s))
    fq, num, fa = b"".join(recs_fq), b"".join(recs_num), b"".join(recs_fa)
    return {"crlf": fq.replace(b"\n", b"\r\n"), "nofinalnl": fq[:-1], "numeric": num, "fasta": fa, "fasta_crlf_nonl": fa.replace(b"\n", b"\r\n")[:-2]}


def test_reader_rules_through_the_batch_path(tools):
    rng = np.random.default_rng(77)
    ad = "AGATCGGAAGAGC"
    for trial in range(4):
        inputs = _odd_inputs(rng)
        for kind, data in inputs.items():
            fasta = kind.startswith("fasta")
            argvs = [["fastx_trimmer", "-f", "3", "-l", "20"], ["fastx_trimmer", "-t", "4", "-m", "5"], ["fastx_reverse_complement"],
                     ["fastx_clipper", "-a", ad, "-l", "5", "-n", "-v"], ["fastx_artifacts_filter", "-v"]]
            if not fasta:
                argvs += [["fastq_quality_trimmer", "-t", "18", "-l", "8", "-v"], ["fastq_quality_filter", "-q", "15", "-p", "60", "-v"],
                          ["fastq_masker", "-q", "12"], ["fastq_to_fasta", "-r"], ["fastx_quality_stats"]]
            for argv in argvs:
                # the device text path (fxg_text.h's per-thread bodies through the emulator), and the host parser on the same input
                for no_text in (False, True):
                    rc, out, err = _run([os.path.join(tools, argv[0])] + argv[1:], data, threads=str([3, 1, 8][trial % 3]), buf_mb="1" if trial % 2 else None,
                                        extra_env={"FXH_TIMING": "1", **({"FXG_EMU_NO_TEXT": "1"} if no_text else {})})
                    rrc, rout, rerr = _run([REF] + argv, data)
                    assert (rc, out) == (rrc, rout), (trial, kind, argv, no_text, err[-200:], rerr[-200:])
                    err = b"".join(l for l in err.splitlines(True) if not l.startswith(b"fxh timing"))
                    assert _msg(err) == _msg(rerr), (trial, kind, argv, no_text)


def test_device_text_path_takes_crlf_numeric_and_fasta_without_the_host_parser(tools):
    """CPU-tier twin of the GPU test: CRLF, missing final newline, numeric quality lines, FASTA with collapsed ids are indexed, packed
    and formatted by the device code (here: its per-thread bodies, run serially) -- no block may fall back to the host parser."""
    rng = np.random.default_rng(79)
    for kind, data in _odd_inputs(rng).items():
        fasta = kind.startswith("fasta")
        argvs = [["fastx_trimmer", "-f", "3", "-l", "20"], ["fastx_reverse_complement"], ["fastx_artifacts_filter", "-v"]]
        if not fasta:
            argvs += [["fastq_quality_trimmer", "-t", "18", "-l", "8", "-v"], ["fastq_masker", "-q", "12"], ["fastq_to_fasta"]]
        for argv in argvs:
            rc, out, err = _run([os.path.join(tools, argv[0])] + argv[1:], data, buf_mb="1", extra_env={"FXH_TIMING": "1"})
            assert rc == 0 and b"device parse" in err and b" 0 host-parsed blocks" in err, (kind, argv, err[-300:])
            if REF:
                assert out == _run([REF] + argv, data)[1], (kind, argv)


def test_sharded_run_parts_concatenate_to_the_single_stream_output(tools, tmp_path):
    """FXH_PARTS=k: k byte ranges of the input cut at record boundaries, k runs side by side (own reader threads, lanes and writer), k
    output parts.  cat(parts) must be the unsharded output, the -v report the same, the index consistent; with two fake devices the
    parts go to different GPUs.  Irregular input anywhere makes the process start over unsharded: messages, exit code and partial
    output are those of the single stream."""
    text = fo.synth_fastq(47, 0, 90000, 100, False)                     # ~20 MB
    inp = tmp_path / "in.fq"
    inp.write_bytes(text)
    argv = ["fastq_quality_trim_filter", "-t", "20", "-l", "30", "-q", "20", "-p", "80", "-v"]
    single = tmp_path / "single.fq"
    want = _run([os.path.join(tools, argv[0])] + argv[1:] + ["-i", str(inp), "-o", str(single)], b"", buf_mb="1")
    assert want[0] == 0 and single.stat().st_size > 5_000_000
    for k, name in ((2, "p.%r.fq"), (3, "q.fq"), (5, "r.%r")):
        pat = str(tmp_path / name)
        log = tmp_path / ("ctx%d.log" % k)
        got = _run([os.path.join(tools, argv[0])] + argv[1:] + ["-i", str(inp), "-o", pat], b"", buf_mb="1",
                   extra_env={"FXH_PARTS": str(k), "FXH_TIMING": "1", "FXG_EMU_DEVICES": "2", "FXG_DEVICES": "0,1", "FXG_EMU_LOG": str(log)})
        assert got[0] == 0 and got[1] == want[1], (k, got[2][-300:])               # the report (stdout with -o) is the single run's
        assert got[2].count(b"fxh timing part") == k and b" 0 host-parsed blocks" in got[2]
        parts = [pat.replace("%r", str(r)) if "%r" in pat else (pat if r == 0 else "%s.%d" % (pat, r)) for r in range(k)]
        assert b"".join(open(f, "rb").read() for f in parts) == single.read_bytes(), k
        ix = open(pat.replace("%r", "parts") if "%r" in pat else pat + ".parts").read().splitlines()[1:]
        assert len(ix) == k and sum(int(l.split("\t")[2]) for l in ix) == len(text) and sum(int(l.split("\t")[3]) for l in ix) == 90000
        assert [int(l.split("\t")[5]) for l in ix] == [os.path.getsize(f) for f in parts]
        devs = [int(l.split()[-1]) for l in open(log).read().splitlines()]
        assert set(devs) == {0, 1}, devs                                           # part r runs on GPU r mod 2
    # irregular input: a damaged record inside the third of four parts, a ragged end, a FASTA/FASTQ mix-up -> the single-stream behaviour
    k0 = text.index(b"\n@", int(len(text) * 0.6)) + 1
    for bad in (text[:k0] + b"#" + text[k0 + 1:], text[:-200], text[:k0] + b"@x\nACGT\n+\nII\n" + text[k0:]):
        inp.write_bytes(bad)
        ref_out = tmp_path / "bad_single.fq"
        w = _run([os.path.join(tools, argv[0])] + argv[1:] + ["-i", str(inp), "-o", str(ref_out)], b"", buf_mb="1")
        pat = str(tmp_path / "bad.%r.fq")
        g = _run([os.path.join(tools, argv[0])] + argv[1:] + ["-i", str(inp), "-o", pat], b"", buf_mb="1", extra_env={"FXH_PARTS": "4"})
        assert w[0] == 1 and (g[0], g[1]) == (w[0], w[1]) and _msg(g[2]) == _msg(w[2]), (g[2][-300:], w[2][-300:])
        assert b"".join(open(pat.replace("%r", str(r)), "rb").read() for r in range(4)) == ref_out.read_bytes()
    # quality lines that start with '@' right where a cut is looked for: the pattern may pick a wrong line, the line count finds out
    recs = [b"@r%d\nACGTACGTACGTACGTACGT\n+\n@IIIIIIIIIIIIIIIIIII\n" % i for i in range(200000)]
    tricky = b"".join(recs)
    inp.write_bytes(tricky)
    w = _run([os.path.join(tools, "fastq_quality_trimmer"), "-t", "20", "-l", "5", "-i", str(inp), "-o", str(tmp_path / "t_single.fq")], b"", buf_mb="1")
    g = _run([os.path.join(tools, "fastq_quality_trimmer"), "-t", "20", "-l", "5", "-i", str(inp), "-o", str(tmp_path / "t.%r.fq")], b"", buf_mb="1", extra_env={"FXH_PARTS": "3", "FXH_TIMING": "1"})
    assert w[0] == 0 and g[0] == 0 and g[2].count(b"fxh timing part") == 3
    assert b"".join(open(str(tmp_path / ("t.%d.fq" % r)), "rb").read() for r in range(3)) == (tmp_path / "t_single.fq").read_bytes()
    # an input too small to shard (and a pipe): one stream, the other parts exist and are empty
    small = fo.synth_fastq(48, 0, 500, 100, False)
    inp.write_bytes(small)
    g = _run([os.path.join(tools, "fastq_quality_trimmer"), "-t", "20", "-l", "5", "-i", str(inp), "-o", str(tmp_path / "s.%r.fq")], b"", extra_env={"FXH_PARTS": "3"})
    assert g[0] == 0 and os.path.getsize(str(tmp_path / "s.0.fq")) > 0 and os.path.getsize(str(tmp_path / "s.1.fq")) == 0 and os.path.getsize(str(tmp_path / "s.2.fq")) == 0
    g = _run([os.path.join(tools, "fastq_quality_trimmer"), "-t", "20", "-l", "5", "-o", str(tmp_path / "pipe.fq")], small, extra_env={"FXH_PARTS": "2"})
    assert g[0] == 0 and (tmp_path / "pipe.fq").read_bytes() == (tmp_path / "s.0.fq").read_bytes() and os.path.getsize(str(tmp_path / "pipe.fq.1")) == 0
    # outputs that cannot take parts: /dev/null, and a sibling name that cannot be created -> one stream, exit code 0, nothing half done
    inp.write_bytes(text)
    g = _run([os.path.join(tools, argv[0])] + argv[1:] + ["-i", str(inp), "-o", "/dev/null"], b"", buf_mb="1", extra_env={"FXH_PARTS": "3", "FXH_TIMING": "1"})
    assert g[0] == 0 and g[1] == want[1] and g[2].count(b"fxh timing part 0/1") == 1, g[2][-300:]
    os.mkdir(str(tmp_path / "blocked.fq.2"))                       # part 2's name is taken by a directory
    g = _run([os.path.join(tools, argv[0])] + argv[1:] + ["-i", str(inp), "-o", str(tmp_path / "blocked.fq")], b"", buf_mb="1", extra_env={"FXH_PARTS": "3", "FXH_TIMING": "1"})
    assert g[0] == 0 and g[1] == want[1] and b"cannot be an output part, running as one stream" in g[2] and g[2].count(b"fxh timing part 0/1") == 1
    assert (tmp_path / "blocked.fq").read_bytes() == single.read_bytes()
    # `-o out.%r.fq` without FXH_PARTS: the tool picks the number of parts by the size of the input (four from FXH_AUTO_PARTS_MIN_MB on, else one)
    for min_mb, nparts in (("1", 4), (None, 1)):
        pat = str(tmp_path / ("auto%s.%%r.fq" % (min_mb or "x")))
        g = _run([os.path.join(tools, argv[0])] + argv[1:] + ["-i", str(inp), "-o", pat], b"", buf_mb="1", extra_env=dict({"FXH_TIMING": "1"}, **({"FXH_AUTO_PARTS_MIN_MB": min_mb} if min_mb else {})))
        assert g[0] == 0 and g[1] == want[1] and g[2].count(b"fxh timing part") == nparts
        assert b"".join(open(pat.replace("%r", str(r)), "rb").read() for r in range(nparts)) == single.read_bytes()
        assert not os.path.exists(pat.replace("%r", str(nparts)))
    # FASTA in, sharded
    fa = b"".join(b">%d-%d\n%s\n" % (i, 1 + i % 7, b"ACGTTGCANN"[: 4 + i % 7] * 3) for i in range(200000))
    inp.write_bytes(fa)
    w = _run([os.path.join(tools, "fastx_reverse_complement"), "-v", "-i", str(inp), "-o", str(tmp_path / "fa_single.fa")], b"", buf_mb="1")
    g = _run([os.path.join(tools, "fastx_reverse_complement"), "-v", "-i", str(inp), "-o", str(tmp_path / "fa.%r.fa")], b"", buf_mb="1", extra_env={"FXH_PARTS": "4"})
    assert w[0] == 0 and (g[0], g[1]) == (w[0], w[1])
    assert b"".join(open(str(tmp_path / ("fa.%d.fa" % r)), "rb").read() for r in range(4)) == (tmp_path / "fa_single.fa").read_bytes()


@pytest.mark.skipif(REF is None, reason="oracle/_ref/fxref not built")
def test_clipper_goes_parallel_while_exact_and_serial_where_it_must(tools, tmp_path):
    """fastx_clipper / fastx_clip_trim_filter with NO environment variable (round-3 verdict item 4): lanes (and parts) work in parallel while
    every block so far holds reads of ONE length -- the reference aligner's stale query tail (sequence_alignment.cpp:135-136)
    only exists once a read shorter than the longest so far turns up -- and the run becomes the reference's one aligner, seeded with the
    last record before, at the first block that is different.  Byte-identical to the real libfastx clipper whether the ragged reads
    start at 0 %, 50 %, 99 % of the file or never, for any lane count, block size and device count; a sharded attempt that meets them
    starts over as one stream."""
    rng = np.random.default_rng(77)
    ad = "AGATCGGAAGAGC"
    fixed = fo.synth_fastq(61, 0, 24000, 100, True).split(b"\n")           # 24 000 records of 100 bases, adapters planted
    recs = [b"\n".join(fixed[4 * i:4 * i + 4]) + b"\n" for i in range(24000)]

    def ragged(rec):
        L = int(rng.integers(20, 100))
        l = rec.split(b"\n")
        return b"\n".join([l[0], l[1][:L], l[2], l[3][:L]]) + b"\n"
    inputs = {}
    for name, start in (("never", None), ("at_0", 0.0), ("at_50", 0.5), ("at_99", 0.99)):
        k = len(recs) if start is None else int(len(recs) * start)
        inputs[name] = b"".join(recs[:k]) + b"".join(ragged(r) if rng.random() < 0.5 else r for r in recs[k:])
    # one shorter block in the middle, fixed (but of ANOTHER length) afterwards: still "a shorter read after a longer one"
    inputs["other_length_later"] = b"".join(recs[:12000]) + b"".join(b"\n".join([r.split(b"\n")[0], r.split(b"\n")[1][:80], b"+", r.split(b"\n")[3][:80]]) + b"\n" for r in recs[12000:])
    for name, data in inputs.items():
        inp = tmp_path / (name + ".fq")
        inp.write_bytes(data)
        for argv in (["fastx_clipper", "-a", ad, "-l", "15", "-v"], ["fastx_clipper", "-a", ad, "-l", "15", "-n", "-c", "-v"]):
            ref = _run([REF] + argv, data)
            assert ref[0] == 0
            for env in ({}, {"FXH_LANES": "3"}, {"FXH_LANES": "2", "FXG_EMU_DEVICES": "2", "FXG_DEVICES": "0,1"}, {"FXH_CLIP_SERIAL": "1"}):
                got = _run([os.path.join(tools, argv[0])] + argv[1:], data, buf_mb="1", extra_env=dict(env, FXH_TIMING="1"))
                assert (got[0], got[1]) == (0, ref[1]), (name, argv, env, got[2][-400:])
                assert [l for l in got[2].splitlines() if not l.startswith(b"fxh ")] == ref[2].splitlines(), (name, env)     # the -v report
                went_serial = b"one aligner with history from there on" in got[2]
                if "FXH_CLIP_SERIAL" in env:
                    assert not went_serial and b" 1 lanes on" in got[2]
                else:
                    assert went_serial == (name != "never"), (name, env, got[2][-300:])
                    assert (b" 1 lanes on" not in got[2]), got[2][-300:]                  # the lanes were there from the start
        # file to file with parts: a sharded attempt is only kept when every part saw the same single length
        out1 = tmp_path / (name + ".single.fq")
        w = _run([os.path.join(tools, "fastx_clipper"), "-a", ad, "-l", "15", "-v", "-i", str(inp), "-o", str(out1)], b"", buf_mb="1")
        pat = str(tmp_path / (name + ".%r.fq"))
        g = _run([os.path.join(tools, "fastx_clipper"), "-a", ad, "-l", "15", "-v", "-i", str(inp), "-o", pat], b"", buf_mb="1", extra_env={"FXH_PARTS": "3", "FXH_TIMING": "1"})
        assert w[0] == 0 and (g[0], g[1]) == (0, w[1]) and out1.read_bytes() == _run([REF, "fastx_clipper", "-a", ad, "-l", "15"], data)[1]
        assert b"".join(open(pat.replace("%r", str(r)), "rb").read() for r in range(3)) == out1.read_bytes(), name
        # kept sharded only for the all-fixed input: an abandoned attempt says so and its rerun is "part 0/1" (counting the parts' timing lines is no
        # test -- an abandoned child prints as many of them as parts had finished)
        assert ((b"parts: abandoned" not in g[2]) and (b"fxh timing part 0/1" not in g[2])) == (name == "never"), (name, g[2][-300:])
    # the one-pass pipeline tool follows the same rule
    data = inputs["at_50"]
    pipe = _run([REF, "fastq_quality_filter", "-q", "20", "-p", "80"], _run([REF, "fastq_quality_trimmer", "-t", "20", "-l", "30"], _run([REF, "fastx_clipper", "-a", ad, "-l", "15", "-n"], data)[1])[1])
    got = _run([os.path.join(tools, "fastx_clip_trim_filter"), "-a", ad, "-l", "15", "-n", "-t", "20", "-m", "30", "-q", "20", "-p", "80"], data, buf_mb="1", extra_env={"FXH_TIMING": "1"})
    assert got[0] == 0 and got[1] == pipe[1] and b"one aligner with history from there on" in got[2]


def test_numa_binding_walks_and_changes_nothing(tools):
    """A run on one GPU binds itself to the CPUs of that GPU's NUMA node (fxh_bind_near_device); the stub has no GPU, FXG_EMU_NUMA_NODE
    names a node so that the code runs: same bytes, also sharded, also when the node does not exist or FXH_NO_NUMA is set."""
    text = fo.synth_fastq(9, 0, 4000, 100, True)
    argv = [os.path.join(tools, "fastq_quality_trim_filter"), "-t", "20", "-l", "30", "-q", "20", "-p", "80", "-v"]
    want = _run(argv, text)
    for env in ({"FXG_EMU_NUMA_NODE": "0"}, {"FXG_EMU_NUMA_NODE": "0", "FXH_LANES": "2", "FXH_READ_BUFFER_MB": "1"}, {"FXG_EMU_NUMA_NODE": "977"},
                {"FXG_EMU_NUMA_NODE": "0", "FXH_NO_NUMA": "1"}):
        got = _run(argv, text, extra_env=env)
        assert got == want, env


def test_lanes_many_blocks_any_lane_count_same_bytes(tools):
    """The lanes loop of the tools (blocks cut at record boundaries on the host, dealt round-robin to lanes over FXG_DEVICES, collected
    in input order): output, report and error behaviour must not depend on the number of lanes, devices or on the block size -- through
    the emulated device text path, through the host parser (FXH_HOST_PARSE) and with every block handed back to it (FXG_EMU_NO_TEXT);
    the GPU tier runs the same matrix against the real engine."""
    text = fo.synth_fastq(41, 0, 30000, 100, False)                     # ~7 MB: seven blocks of 1 MB
    argv = ["fastq_quality_trimmer", "-t", "20", "-l", "30", "-v"]
    base = _run([os.path.join(tools, argv[0])] + argv[1:], text)
    assert base[0] == 0
    for env in ({"FXH_LANES": "1"}, {"FXH_LANES": "3"}, {"FXG_DEVICES": "0,0,0", "FXH_LANES": "2"}, {"FXH_NO_OVERLAP": "1"}, {"FXH_HOST_PARSE": "1"},
                {"FXG_EMU_NO_TEXT": "1", "FXH_LANES": "3"}, {"FXG_EMU_DEVICES": "3", "FXG_DEVICES": "0,2,1", "FXH_LANES": "2"}):
        for buf in ("1", "2", None):
            assert _run([os.path.join(tools, argv[0])] + argv[1:], text, buf_mb=buf, extra_env=env) == base, (env, buf)
    # a damaged record in the fifth megabyte: everything before it is written, then the reference's message and exit 1
    k = text.index(b"\n@", 4_500_000) + 1
    bad = text[:k] + b"#" + text[k + 1:]
    want = _run([os.path.join(tools, argv[0])] + argv[1:], bad, extra_env={"FXH_HOST_PARSE": "1", "FXH_NO_OVERLAP": "1"})
    assert want[0] == 1 and len(want[1]) > 1_000_000
    if REF:
        ref = _run([REF] + argv, bad)
        assert (want[0], want[1]) == (ref[0], ref[1]) and _msg(want[2]) == _msg(ref[2])
    for env in ({"FXH_LANES": "1"}, {"FXH_LANES": "4"}, {"FXG_DEVICES": "0,0", "FXH_LANES": "2"}):
        got = _run([os.path.join(tools, argv[0])] + argv[1:], bad, buf_mb="1", extra_env=env)
        assert (got[0], got[1]) == (want[0], want[1]) and _msg(got[2]) == _msg(want[2]), env
    # ragged end of input (a record cut short) and input without a final newline
    for tail in (text[:-1], text[:-150], text + b"@x\nAC\n"):
        want = _run([os.path.join(tools, argv[0])] + argv[1:], tail, extra_env={"FXH_HOST_PARSE": "1"})
        for env in ({"FXH_LANES": "1"}, {"FXH_LANES": "3"}):
            got = _run([os.path.join(tools, argv[0])] + argv[1:], tail, buf_mb="1", extra_env=env)
            assert (got[0], got[1]) == (want[0], want[1]) and _msg(got[2]) == _msg(want[2]), env


def test_fused_trim_filter_equals_the_pipe(tools):
    """fastq_quality_trim_filter (one pass) writes the bytes of fastq_quality_trimmer | fastq_quality_filter, and -v prints both reports."""
    text = fo.synth_fastq(2, 0, 4000, 150, False)
    t = _run([os.path.join(tools, "fastq_quality_trimmer"), "-t", "20", "-l", "30", "-v"], text)
    f = _run([os.path.join(tools, "fastq_quality_filter"), "-q", "20", "-p", "80", "-v"], t[1])
    one = _run([os.path.join(tools, "fastq_quality_trim_filter"), "-t", "20", "-l", "30", "-q", "20", "-p", "80", "-v"], text)
    assert one[0] == 0 and one[1] == f[1]
    assert one[2] == t[2] + f[2]
    if REF:
        rt = _run([REF, "fastq_quality_trimmer", "-t", "20", "-l", "30", "-v"], text)
        rf = _run([REF, "fastq_quality_filter", "-q", "20", "-p", "80", "-v"], rt[1])
        assert one[1] == rf[1] and one[2] == rt[2] + rf[2]
    assert _run([os.path.join(tools, "fastq_quality_trim_filter"), "-q", "20"], text)[0] == 1            # -t is mandatory, as for the trimmer


def test_fused_clip_trim_filter_equals_the_three_tool_pipe(tools):
    """fastx_clip_trim_filter (BASELINE config 5 in one pass) writes the bytes of fastx_clipper | fastq_quality_trimmer | fastq_quality_filter
    and prints the three reports in turn -- against the real libfastx driver, fixed-length and ragged input, several flag sets."""
    rng = np.random.default_rng(31)
    ad = "AGATCGGAAGAGC"
    from helpers import random_batch
    texts = [fo.synth_fastq(5, 0, 3000, 150, True)]
    b, q, lens = random_batch(rng, 800, 90, 20, 90, False, adapter=ad.encode())
    texts.append(b"".join(b"@v%d\n%s\n+\n%s\n" % (i, bytes(b[i, :lens[i]]), bytes(q[i, :lens[i]])) for i in range(800)))
    for text in texts:
        for cflags, extra in (([], []), (["-n"], []), (["-c"], ["-m", "12"]), (["-C", "-n", "-M", "6"], ["-m", "5"])):
            c = _run([os.path.join(tools, "fastx_clipper"), "-a", ad, "-l", "15", "-v"] + cflags, text)
            t = _run([os.path.join(tools, "fastq_quality_trimmer"), "-t", "20", "-v"] + (["-l", extra[1]] if extra else []), c[1])
            f = _run([os.path.join(tools, "fastq_quality_filter"), "-q", "20", "-p", "80", "-v"], t[1])
            one = _run([os.path.join(tools, "fastx_clip_trim_filter"), "-a", ad, "-l", "15", "-t", "20", "-q", "20", "-p", "80", "-v"] + cflags + extra, text)
            assert one[0] == 0 and one[1] == f[1], (cflags, extra, one[2][-300:])
            assert one[2] == c[2] + t[2] + f[2], (cflags, extra, one[2], c[2] + t[2] + f[2])
            if REF:
                rc_ = _run([REF, "fastx_clipper", "-a", ad, "-l", "15", "-v"] + cflags, text)
                rt = _run([REF, "fastq_quality_trimmer", "-t", "20", "-v"] + (["-l", extra[1]] if extra else []), rc_[1])
                rf = _run([REF, "fastq_quality_filter", "-q", "20", "-p", "80", "-v"], rt[1])
                assert one[1] == rf[1] and one[2] == rc_[2] + rt[2] + rf[2], (cflags, extra)
    assert _run([os.path.join(tools, "fastx_clip_trim_filter"), "-q", "20"], texts[0])[0] == 1            # -t is mandatory, as for the trimmer


def test_regular_files_use_parallel_io_same_bytes(tools, tmp_path):
    """-i FILE / -o FILE: blocks are read with several pread() in flight and written with positional writes; same bytes as through pipes."""
    text = fo.synth_fastq(43, 0, 120000, 100, False)                    # ~28 MB
    argv = ["fastq_quality_trimmer", "-t", "20", "-l", "30"]
    want = _run([os.path.join(tools, argv[0])] + argv[1:], text)
    assert want[0] == 0
    inp, outp = tmp_path / "in.fq", tmp_path / "out.fq"
    inp.write_bytes(text)
    for buf, io in (("24", "5"), ("24", "1"), ("9", "2"), (None, "8")):
        rc, out, err = _run([os.path.join(tools, argv[0])] + argv[1:] + ["-i", str(inp), "-o", str(outp)], b"", buf_mb=buf, extra_env={"FXH_IO_THREADS": io})
        assert rc == 0 and outp.read_bytes() == want[1], (buf, io)
    with open(outp, "wb") as f:                                         # a descriptor that is not at offset 0, and one in append mode
        f.write(b"HEAD\n")
    with open(outp, "ab") as f:
        p = subprocess.run([os.path.join(tools, argv[0])] + argv[1:] + ["-i", str(inp)], stdout=f, env=dict(os.environ, LD_LIBRARY_PATH=STUB_DIR, FXH_READ_BUFFER_MB="24"), timeout=120)
    assert p.returncode == 0 and outp.read_bytes() == b"HEAD\n" + want[1]


@pytest.mark.skipif(REF is None, reason="oracle/_ref/fxref not built")
def test_clipper_long_fixed_length_reads_get_dword_rows(tools):
    """fastx_clipper / fastx_clip_trim_filter on reads of ONE length beyond 160 bases: the lanes pack the rows with a stride rounded up to a
    multiple of four (host/fxh_lanes.c) so that the clip kernel may read them where they are (csrc/fxg_plan.h: clip_global).  Lengths that are
    and are not multiples of four, with and without -n, against the real libfastx."""
    ad = "AGATCGGAAGAGC"
    for L in (161, 250, 251, 300, 301):
        data = fo.synth_fastq(70 + L, 0, 3000, L, True)
        for argv in (["fastx_clipper", "-a", ad, "-l", "15", "-v"], ["fastx_clipper", "-a", ad, "-l", "15", "-n", "-v"]):
            ref = _run([REF] + argv, data)
            got = _run([os.path.join(tools, argv[0])] + argv[1:], data, extra_env={"FXH_TIMING": "1"})
            assert (got[0], got[1]) == (0, ref[1]), (L, argv, got[2][-300:])
            assert b" 0 host-parsed blocks" in got[2], got[2][-300:]
        chain = [["fastx_clipper", "-a", ad, "-l", "15", "-n"], ["fastq_quality_trimmer", "-t", "20", "-l", "30"], ["fastq_quality_filter", "-q", "20", "-p", "80"]]
        want = data
        for c in chain:
            want = _run([REF] + c, want)[1]
        got = _run([os.path.join(tools, "fastx_clip_trim_filter"), "-a", ad, "-l", "15", "-n", "-t", "20", "-m", "30", "-q", "20", "-p", "80"], data)
        assert (got[0], got[1]) == (0, want), (L, got[2][-300:])


def test_one_long_read_among_short_ones_does_not_blow_up_the_rows(tools):
    """The SoA rows are n x longest read: the host path cuts a batch at a record boundary when one long read would make the rows
    many times the size of the text (a 20 kb read among 200 k short ones would otherwise ask for gigabytes of pinned memory)."""
    rng = np.random.default_rng(5)
    recs = [b"@s%d\nACGTACGTAC\n+\nIIIIIIIIII\n" % i for i in range(150000)]
    longr = b"@long\n" + rng.choice(np.frombuffer(b"ACGT", np.uint8), size=20000).tobytes() + b"\n+\n" + b"I" * 20000 + b"\n"
    data = b"".join(recs[:75000]) + longr + b"".join(recs[75000:]) + longr
    for argv in (["fastq_quality_trimmer", "-t", "20", "-l", "5", "-v"], ["fastx_reverse_complement"]):
        rc, out, err = _run([os.path.join(tools, argv[0])] + argv[1:], data, extra_env={"FXH_HOST_PARSE": "1", "FXH_TIMING": "1"})
        assert rc == 0, err[-300:]
        if REF:
            rrc, rout, rerr = _run([REF] + argv, data)
            assert (rc, out) == (rrc, rout)
        else:
            assert out.count(b"\n") == (4 * 150002 if argv[0] != "fastq_quality_trimmer" else out.count(b"\n"))


def _corner_corpus():
    """Hand-written malformed / corner-case inputs (the reader rules R1-R9 at their edges): what the reference does with each is
    whatever its fgets + chomp + strlen code does, and the tools must do the same -- exit code, stdout, message."""
    rec = b"@r1\nACGTACGTAC\n+\nIIIIIIIIII\n"
    rec2 = b"@r2 desc\nTTGCANNACG\n+r2 desc\n#5?IIII#5I\n"
    C = {
     "empty": b"", "only_newline": b"\n", "just_at": b"@", "header_only": b"@r1\n", "two_lines": b"@r1\nACGT\n", "three_lines": b"@r1\nACGT\n+\n",
     "no_final_newline": rec[:-1], "truncated_quality": b"@r1\nACGTACGT\n+\nIII\n", "long_quality": b"@r1\nACGT\n+\nIIIIIIII\n",
     "bad_base": b"@r1\nACGXT\n+\nIIIII\n", "lowercase": b"@r1\nacgt\n+\nIIII\n", "empty_sequence": b"@r1\n\n+\n\n", "no_plus": b"@r1\nACGT\n-\nIIII\n",
     "crlf": rec.replace(b"\n", b"\r\n") * 3, "cr_only": rec.replace(b"\n", b"\r"), "mixed_numeric": rec + b"@r2\nACGT\n+\n40 40 30 2\n" + rec,
     "numeric_overflow": b"@r1\nAC\n+\n99999999999999999999 3\n", "numeric_negative": b"@r1\nACG\n+\n-5 -15 -16\n", "numeric_junk": b"@r1\nACG\n+\n3 x 4\n",
     "quality_too_high": b"@r1\nACGT\n+\nII\x7fI\n", "quality_too_low": b"@r1\nACGT\n+\nII\x10I\n", "high_bit": b"@r1\nACGT\n+\nII\xffI\n",
     "fasta": b">1-5\nACGTN\n>2\nTTGCA\n", "fasta_in_fastq": rec + b">x\nACGT\n", 
     "blank_between": rec + b"\n" + rec, "garbage": bytes(range(256)) * 4, "many_small": b"@\nA\n\nI\n" * 50,
     "second_record_bad": rec * 50 + b"@bad\nACGT\n+\nII\n" + rec * 50,
     # new ones
     "leading_blank": b"\n" + rec, "trailing_blanks": rec + b"\n\n", "space_in_seq": b"@r1\nACG T\n+\nIIIII\n", "tab_in_seq": b"@r1\nACG\tT\n+\nIIIII\n",
     "trailing_space_seq": b"@r1\nACGT \n+\nIIII\n", "trailing_space_qual": b"@r1\nACGT\n+\nIIII \n", "numeric_trailing_space": b"@r1\nACGT\n+\n40 40 40 40 \n",
     "numeric_leading_space": b"@r1\nACGT\n+\n 40 40 40 40\n", "numeric_double_space": b"@r1\nACGT\n+\n40  40 40 40\n", "numeric_tabs": b"@r1\nACGT\n+\n40\t40\t40\t40\n",
     "numeric_plus_sign": b"@r1\nACGT\n+\n+40 +4 40 40\n", "numeric_too_few": b"@r1\nACGT\n+\n40 40 40\n", "numeric_too_many": b"@r1\nACGT\n+\n40 40 40 40 40\n",
     "numeric_single_base": b"@r1\nA\n+\n40\n", "ascii_single_digit": b"@r1\nA\n+\n5\n", "ascii_digits_same_len": b"@r1\nACG\n+\n555\n", "ascii_with_space": b"@r1\nACG\n+\nI I\n",
     "numeric_big_in_range": b"@r1\nAC\n+\n93 -40\n", "numeric_out_of_range_hi": b"@r1\nAC\n+\n94 3\n", "numeric_out_of_range_lo": b"@r1\nAC\n+\n-41 3\n",
     "plus_with_other_name": b"@r1\nACGT\n+zzz\nIIII\n", "at_in_quality": b"@r1\nACGT\n+\n@@@@\n" + rec, "gt_first_in_fastq_tool": b">r1\nACGT\n" + rec,
     "fasta_multiline": b">r1\nACGT\nACGT\n>r2\nAC\n", "fasta_empty_seq": b">r1\n\n>r2\nACGT\n", "fasta_no_final_nl": b">r1\nACGT", "fasta_lower": b">r1\nacgt\n", 
     "fasta_blank_lines": b">r1\nACGT\n\n>r2\nACGT\n", "fasta_only_header": b">r1\n", "fasta_N_only": b">r1\nNNNN\n", "fasta_collapsed": b">12-345\nACGTACGT\n>13-1\nACGT\n",
     "fasta_bad_collapsed": b">12-abc\nACGT\n>-5\nAC\n>3-\nAC\n", "nul_byte_in_seq": b"@r1\nAC\x00GT\n+\nIIIII\n", "nul_in_id": b"@r\x001\nACGT\n+\nIIII\n",
     "seq_only_N": b"@r1\nNNNNNN\n+\nIIIIII\n", "one_base": b"@r1\nA\n+\nI\n", "qual_all_low": b"@r1\nACGTACGT\n+\n!!!!!!!!\n", "qual_max": b"@r1\nACGT\n+\n}}}}\n",
     "crlf_numeric": b"@r1\r\nACGT\r\n+\r\n40 40 30 2\r\n", "crlf_no_final": rec.replace(b"\n", b"\r\n")[:-2], "cr_at_end_only": rec[:-1] + b"\r\n",
     "lf_cr": rec.replace(b"\n", b"\n\r"), "dos_eof": rec + b"\x1a", "bom": b"\xef\xbb\xbf" + rec, "id_empty": b"@\nACGT\n+\nIIII\n", "id_spaces": b"@  a  b \nACGT\n+\nIIII\n",
     "two_ok_then_trunc": rec + rec2 + b"@r3\nACGT\n+\n", "two_ok_then_header": rec + rec2 + b"@r3", "ok_then_garbage": rec + b"hello\n", "u_base": b"@r1\nACGU\n+\nIIII\n",
     "iupac": b"@r1\nACGTRYKM\n+\nIIIIIIII\n", "dot_base": b"@r1\nAC.T\n+\nIIII\n", "dash_base": b"@r1\nAC-T\n+\nIIII\n", "star_qual": rec2*3,
     "long_read_2000": b"@r1\n" + b"ACGT"*500 + b"\n+\n" + b"I"*2000 + b"\n", "mixed_lengths": rec + b"@s\nAC\n+\nII\n" + b"@t\n" + b"G"*300 + b"\n+\n" + b"5"*300 + b"\n" + rec2,
     "numeric_then_ascii": b"@r2\nACGT\n+\n40 40 30 2\n" + rec, "numeric_minus_alone": b"@r1\nAC\n+\n- 3\n", "numeric_decimal": b"@r1\nAC\n+\n3.5 3\n", "numeric_hex": b"@r1\nAC\n+\n0x10 3\n",
     "numeric_leading_zero": b"@r1\nAC\n+\n007 010\n", "seq_with_cr_mid": b"@r1\nAC\rGT\n+\nIIIII\n",
    }
    return C


def test_corner_case_inputs_vs_reference(tools):
    """Every corner-case input through the tools -- device text path (emulated) and host parser -- against the real libfastx."""
    if not REF:
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    argvs = [["fastq_quality_trimmer", "-t", "20", "-l", "2", "-v"], ["fastq_quality_filter", "-q", "20", "-p", "50", "-v"], ["fastx_trimmer", "-f", "2", "-l", "9"],
             ["fastx_reverse_complement"], ["fastx_clipper", "-a", "GTAC", "-l", "2", "-v"], ["fastq_masker", "-q", "30"], ["fastq_to_fasta", "-n"],
             ["fastx_artifacts_filter", "-v"], ["fastx_quality_stats"], ["fastx_quality_stats", "-N"], ["fastx_trimmer", "-t", "2", "-m", "3"],
             ["fastx_clipper", "-a", "CGTA", "-l", "1", "-n", "-C"]]
    from concurrent.futures import ThreadPoolExecutor

    def one(job):
        name, data, argv = job
        r = _run([REF] + argv, data)
        for mode in ({}, {"FXH_HOST_PARSE": "1"}):
            g = _run([os.path.join(tools, argv[0])] + argv[1:], data, threads="2", extra_env=mode)
            assert (g[0], g[1]) == (r[0], r[1]) and _msg(g[2]) == _msg(r[2]), (name, argv, mode, g[0], g[1][:80], g[2][:200], r[0], r[1][:80], r[2][:200])
        return 2

    # (collapsed FASTA with hundreds of reads per record into fastx_quality_stats -N: the reference's percentile walk leaves its
    #  25 000-cycle array and reads unrelated globals -- no defined answer to compare with; the tool stops at the array's end)
    jobs = [(name, data, argv) for name, data in _corner_corpus().items() for argv in argvs if not (name == "fasta_collapsed" and argv[-1] == "-N")]
    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 4)) as ex:
        checked = sum(ex.map(one, jobs))
    assert checked > 1500


def _one_file_env(sink, kb="512", strands="3", **more):
    env = {"FXH_ONE_FILE_MIN_MB": "0", "FXH_STRAND_KB": kb, "FXH_STRANDS": strands, "FXH_ONE_FILE_SINK": sink, "FXH_ONE_FILE_WINDOW_MB": "1", "FXH_TIMING": "1"}
    env.update(more)
    return env


@pytest.mark.parametrize("sink", ["map", "pwrite"])
def test_one_output_file_written_by_many_strands(tools, tmp_path, sink):
    """`tool -i in.fq -o out.fq` with NO %r and no FXH_PARTS (fxh_strands.c): chunks of the input dealt to strands by a ticket counter, sizes published
    in chunk order, every chunk's text copied into the ONE output file at the sum of the sizes before it -- through the gated mapping (allocator
    and copies taking turns) and through positional writes.  Same bytes, same -v report as the one-stream run for every tool shape (forward
    slices, reverse-complemented / masked output from the packed arrays, FASTA out, FASTA in with collapsed ids, the clipper on reads of one
    length); two fake devices share the strands; input on stdin as a regular file works, a pipe runs as one stream."""
    text = fo.synth_fastq(47, 0, 60000, 100, False)                     # ~13 MB: 26 chunks of 512 KB
    inp = tmp_path / "in.fq"
    inp.write_bytes(text)
    clip_text = fo.synth_fastq(3, 0, 30000, 100, True)
    fa = b"".join(b">%d-%d\n%s\n" % (i, 1 + i % 7, b"ACGTTGCANN"[: 4 + i % 7] * 3) for i in range(120000))
    shapes = [(["fastq_quality_trim_filter", "-t", "20", "-l", "30", "-q", "20", "-p", "80", "-v"], text),
              (["fastx_reverse_complement", "-v"], text), (["fastq_masker", "-q", "20", "-v"], text), (["fastq_to_fasta", "-v"], text),
              (["fastx_trimmer", "-f", "5", "-l", "80"], text), (["fastx_clipper", "-a", "AGATCGGAAGAGC", "-l", "15", "-v"], clip_text),
              (["fastx_clip_trim_filter", "-a", "AGATCGGAAGAGC", "-l", "15", "-t", "20", "-m", "30", "-q", "20", "-p", "80", "-v"], clip_text),
              (["fastx_reverse_complement", "-v"], fa), (["fastx_artifacts_filter", "-v"], fa)]
    for i, (argv, data) in enumerate(shapes):
        inp.write_bytes(data)
        single, multi = tmp_path / ("single%d" % i), tmp_path / ("multi%d" % i)
        want = _run([os.path.join(tools, argv[0])] + argv[1:] + ["-i", str(inp), "-o", str(single)], b"", buf_mb="1", extra_env={"FXH_ONE_FILE": "0"})
        log = tmp_path / ("ctx%d.log" % i)
        got = _run([os.path.join(tools, argv[0])] + argv[1:] + ["-i", str(inp), "-o", str(multi)], b"", buf_mb="1",
                   extra_env=_one_file_env(sink, FXG_EMU_DEVICES="2", FXG_DEVICES="0,1", FXG_EMU_LOG=str(log)))
        assert want[0] == 0 and got[0] == 0, (argv, got[2][-400:])
        assert b"fxh timing one file (6 strands on 2 GPU(s)" in got[2] and (b"sink gated mapping" if sink == "map" else b"sink pwrite") in got[2], got[2][-400:]
        assert b"fxh timing part" not in got[2]                         # the one-stream loop never ran
        assert got[1] == want[1], argv                                  # the -v report
        assert multi.read_bytes() == single.read_bytes(), argv
        assert {int(l.split()[-1]) for l in open(log).read().splitlines()} == {0, 1}
    inp.write_bytes(text)
    argv = shapes[0][0]
    single = (tmp_path / "single0").read_bytes()
    # the input as stdin, a regular file: the same run; as a pipe: one stream, the same bytes
    with open(inp, "rb") as f:
        env = dict(os.environ, LD_LIBRARY_PATH=STUB_DIR, FXH_THREADS="4", **_one_file_env(sink))
        p = subprocess.run([os.path.join(tools, argv[0])] + argv[1:] + ["-o", str(tmp_path / "stdin.fq")], stdin=f, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=120)
    assert p.returncode == 0 and b"fxh timing one file (3 strands" in p.stderr and (tmp_path / "stdin.fq").read_bytes() == single
    g = _run([os.path.join(tools, argv[0])] + argv[1:] + ["-o", str(tmp_path / "pipe1.fq")], text, extra_env=_one_file_env(sink))
    assert g[0] == 0 and b"fxh timing one file" not in g[2] and (tmp_path / "pipe1.fq").read_bytes() == single
    # the rank path with a world of one (what the GPU tier runs against the real RCCL): arena, one all-gather over the fake library, drain at offset 0
    import emu_py
    out = tmp_path / "rank_mode_world1.fq"
    g = _run([os.path.join(tools, argv[0])] + argv[1:] + ["-i", str(inp), "-o", str(out)], b"",
             extra_env=dict(_one_file_env(sink), FXH_RANK_MODE="1", LD_LIBRARY_PATH=STUB_DIR + os.pathsep + emu_py.build_fake_rccl()))
    assert g[0] == 0 and b"fxh timing rank 0 of 1" in g[2] and out.read_bytes() == single, g[2][-300:]
    # one strand, many strands, chunks of 64 KB (800 of them) and of 5 MB (three); more strands than chunks
    for kb, strands in (("64", "1"), ("64", "8"), ("5120", "4"), ("7000", "16")):
        out = tmp_path / ("o_%s_%s.fq" % (kb, strands))
        g = _run([os.path.join(tools, argv[0])] + argv[1:] + ["-i", str(inp), "-o", str(out)], b"", extra_env=_one_file_env(sink, kb=kb, strands=strands))
        assert g[0] == 0 and b"fxh timing one file" in g[2] and out.read_bytes() == single, (kb, strands, g[2][-300:])
    # below the size from which it pays (the default: 1 GB) the run is one stream; -z, /dev/null and FXH_ONE_FILE=0 as well
    for extra, args in (({"FXH_TIMING": "1"}, []), (dict(_one_file_env(sink), FXH_ONE_FILE="0"), []), (_one_file_env(sink), ["-z"])):
        out = tmp_path / "plain.out"
        g = _run([os.path.join(tools, argv[0])] + argv[1:] + args + ["-i", str(inp), "-o", str(out)], b"", extra_env=extra)
        assert g[0] == 0 and b"fxh timing one file" not in g[2] and b"fxh timing part 0/1" in g[2]
    g = _run([os.path.join(tools, argv[0])] + argv[1:] + ["-i", str(inp), "-o", "/dev/null"], b"", extra_env=_one_file_env(sink))
    assert g[0] == 0 and b"fxh timing one file" not in g[2]
    if sink == "map":
        # a tool that keeps little: the allocator's head start (a quarter of the input, made before any chunk is back) is several times the output; once the measured
        # ratio says so the surplus pages go back during the run (a hole punched above estimate + window), not in the ftruncate() at its end.  Same bytes.
        fargv = ["fastq_quality_filter", "-q", "39", "-p", "60", "-v"]
        want = _run([os.path.join(tools, fargv[0])] + fargv[1:] + ["-i", str(inp), "-o", str(tmp_path / "light_single.fq")], b"", extra_env={"FXH_ONE_FILE": "0"})
        g = _run([os.path.join(tools, fargv[0])] + fargv[1:] + ["-i", str(inp), "-o", str(tmp_path / "light.fq")], b"", extra_env=_one_file_env(sink, FXH_ONE_FILE_RATIO_MB="0"))
        assert want[0] == 0 and g[0] == 0 and g[1] == want[1] and (tmp_path / "light.fq").read_bytes() == (tmp_path / "light_single.fq").read_bytes()
        assert len((tmp_path / "light.fq").read_bytes()) < len(text) // 20
        line = [l for l in g[2].decode().splitlines() if l.startswith("fxh timing one file (")][0]
        assert float(line.split(" GB of output, ")[1].split(" MB given back")[0]) >= 1.0, line


@pytest.mark.parametrize("sink", ["map", "pwrite"])
def test_one_output_file_attempt_abandoned_on_irregular_input(tools, tmp_path, sink):
    """Anything the device path does not take -- a damaged record, a ragged end, a record of another format, a clipper input whose reads stop being of
    one length, quality lines starting with '@' where the cuts are looked for -- abandons the many-strand attempt: the file is emptied and the
    input runs as one stream, so exit code, message and the bytes written before the bad record are the reference's."""
    text = fo.synth_fastq(47, 0, 60000, 100, False)
    inp = tmp_path / "in.fq"
    argv = ["fastq_quality_trim_filter", "-t", "20", "-l", "30", "-q", "20", "-p", "80", "-v"]
    k0 = text.index(b"\n@", int(len(text) * 0.6)) + 1
    for n, bad in enumerate((text[:k0] + b"#" + text[k0 + 1:], text[:-200], text[:k0] + b">x\nACGT\n" + text[k0:], text[:k0] + b"@x\nACGT\n+\nII\n" + text[k0:])):
        inp.write_bytes(bad)
        ref_out, out = tmp_path / ("bad_single%d.fq" % n), tmp_path / ("bad%d.fq" % n)
        w = _run([os.path.join(tools, argv[0])] + argv[1:] + ["-i", str(inp), "-o", str(ref_out)], b"", buf_mb="1", extra_env={"FXH_ONE_FILE": "0"})
        g = _run([os.path.join(tools, argv[0])] + argv[1:] + ["-i", str(inp), "-o", str(out)], b"", buf_mb="1", extra_env=_one_file_env(sink))
        if n < 3:
            assert w[0] == 1 and b"fxh timing one file: abandoned, contexts destroyed, output emptied" in g[2], g[2][-300:]
        assert (g[0], g[1]) == (w[0], w[1]) and _msg(g[2].split(b"\n", g[2].count(b"\n") - 1)[-1]) == _msg(w[2].split(b"\n", w[2].count(b"\n") - 1)[-1]), (n, g[2][-300:], w[2][-300:])
        assert out.read_bytes() == ref_out.read_bytes(), n
        if REF and n < 3:
            r1 = _run([REF, "fastq_quality_trimmer", "-t", "20", "-l", "30"], bad)
            assert r1[0] == 1 and out.read_bytes() == _run([REF, "fastq_quality_filter", "-q", "20", "-p", "80"], r1[1])[1]
    # the clipper: reads of one length up to 70 % of the file, shorter ones from there on -> one aligner with history, as the reference
    lines = fo.synth_fastq(3, 0, 30000, 100, True).split(b"\n")[:-1]
    for i in range(len(lines) // 4 * 7 // 10 * 4, len(lines), 8):
        lines[i + 1] = lines[i + 1][:61]; lines[i + 3] = lines[i + 3][:61]
    data = b"\n".join(lines) + b"\n"
    inp.write_bytes(data)
    cargv = ["fastx_clipper", "-a", "AGATCGGAAGAGC", "-l", "15", "-v"]
    g = _run([os.path.join(tools, cargv[0])] + cargv[1:] + ["-i", str(inp), "-o", str(tmp_path / "clip.fq")], b"", buf_mb="1", extra_env=_one_file_env(sink))
    assert g[0] == 0 and b"fxh timing one file: abandoned" in g[2] and b"one aligner with history from there on" in g[2], g[2][-400:]
    if REF:
        ref = _run([REF] + cargv, data)
        assert g[1] == ref[2] and (tmp_path / "clip.fq").read_bytes() == ref[1]
    # quality lines that start with '@': a cut may be found on a wrong line; the chunk before it is then not whole records
    tricky = b"".join(b"@r%d\nACGTACGTACGTACGTACGT\n+\n@IIIIIIIIIIIIIIIIIII\n" % i for i in range(120000))
    inp.write_bytes(tricky)
    w = _run([os.path.join(tools, "fastq_quality_trimmer"), "-t", "20", "-l", "5", "-i", str(inp), "-o", str(tmp_path / "t_single.fq")], b"", buf_mb="1", extra_env={"FXH_ONE_FILE": "0"})
    g = _run([os.path.join(tools, "fastq_quality_trimmer"), "-t", "20", "-l", "5", "-i", str(inp), "-o", str(tmp_path / "t.fq")], b"", buf_mb="1", extra_env=_one_file_env(sink, kb="100"))
    assert w[0] == 0 and g[0] == 0 and (tmp_path / "t.fq").read_bytes() == (tmp_path / "t_single.fq").read_bytes()


def _rank_job(tools, argv, inp, out, world, tmp_path, extra=None, delays=None):
    """One process per rank, the SAME command line, FXH_RANK / FXH_WORLD in the environment (what mpirun, srun or a shell loop would start)."""
    import emu_py
    fake = emu_py.build_fake_rccl()
    procs = []
    for r in range(world):
        env = dict(os.environ, LD_LIBRARY_PATH=STUB_DIR + os.pathsep + fake, FXH_THREADS="2", FXH_RANK=str(r), FXH_WORLD=str(world), FXH_TIMING="1", FXH_STRAND_KB="256",
                   FXH_STRANDS="2", FXH_DRAIN_MB="1", FXG_EMU_DEVICES="2")
        env.update(extra or {})
        cmd = [os.path.join(tools, argv[0])] + argv[1:] + ["-i", str(inp), "-o", str(out)]
        if delays and r in delays:
            cmd = ["sh", "-c", "sleep %s; exec \"$@\"" % delays[r], "sh"] + cmd
        procs.append(subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env))
    res = []
    for p in procs:
        o, e = p.communicate(timeout=180)
        res.append((p.returncode, o, e))
    return res


@pytest.mark.parametrize("world", [2, 3, 8])
def test_rank_per_gpu_job_writes_the_one_file_and_rank0_reports(tools, tmp_path, world):
    """north_star's multi-GPU form at the level users run (round-4 verdict item 2): `world` processes, one per GPU, the same command line each; rank g takes byte
    range g of the input (cut at record starts), keeps its formatted text on its device, the ranks exchange their counter blocks in ONE all-gather
    (the product's fxg_comm_create / fxg_epilogue_rccl, here over tests/emu/fake_rccl.c) and every rank writes its slice of the ONE output file at the
    sum of the bytes of the ranks before it; rank 0 prints the -v report of the whole job.  Bytes and report equal the single-process run's."""
    text = fo.synth_fastq(47, 0, 40000, 100, False)
    clip_text = fo.synth_fastq(3, 0, 20000, 100, True)
    inp = tmp_path / "in.fq"
    for i, (argv, data) in enumerate([(["fastq_quality_trim_filter", "-t", "20", "-l", "30", "-q", "20", "-p", "80", "-v"], text),
                                      (["fastx_reverse_complement", "-v"], text),
                                      (["fastx_clipper", "-a", "AGATCGGAAGAGC", "-l", "15", "-v"], clip_text)]):
        inp.write_bytes(data)
        single, multi = tmp_path / ("single%d" % i), tmp_path / ("ranks%d" % i)
        want = _run([os.path.join(tools, argv[0])] + argv[1:] + ["-i", str(inp), "-o", str(single)], b"", extra_env={"FXH_ONE_FILE": "0"})
        multi.write_bytes(b"stale bytes of an earlier run, longer than nothing\n" * 3)
        res = _rank_job(tools, argv, inp, multi, world, tmp_path, delays={0: 0.3} if i == 1 else None)      # (once with a late rank 0: the others wait at the rendezvous)
        assert want[0] == 0 and all(rc == 0 for rc, _, _ in res), [e[-300:] for _, _, e in res]
        assert multi.read_bytes() == single.read_bytes(), (argv, world)
        assert res[0][1] == want[1] and all(o == b"" for _, o, _ in res[1:])          # the job's report, from rank 0 alone
        lines = [l for _, _, e in res for l in e.decode().splitlines() if l.startswith("fxh timing rank ")]
        assert len(lines) == world
        spans = sorted((int(l.split("[")[1].split(",")[0]), int(l.split(", ")[1].split(")")[0])) for l in lines)
        assert spans[0][0] == 0 and spans[-1][1] == len(data) and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))      # the ranges tile the input
        assert not os.path.exists(str(multi) + ".rdv")
        assert all("positional writes" in l for l in lines)              # (not a tmpfs)
        # the same job with the sink a tmpfs gets: rank 0 makes the JOB's pages while the ranks compute (eager fallocate towards the expected size, exact after the
        # exchange), a third exchange is the barrier, every rank copies into a mapping of its own slice -- no rank allocates under the one inode lock
        paged = tmp_path / ("paged%d" % i)
        paged.write_bytes(b"stale bytes of an earlier run, longer than nothing\n" * 3)
        res = _rank_job(tools, argv, inp, paged, world, tmp_path, extra={"FXH_ONE_FILE_SINK": "map", "FXH_ONE_FILE_WINDOW_MB": "1"})
        assert all(rc == 0 for rc, _, _ in res), [e[-300:] for _, _, e in res]
        assert paged.read_bytes() == single.read_bytes() and res[0][1] == want[1], (argv, world)
        lines = [l for _, _, e in res for l in e.decode().splitlines() if l.startswith("fxh timing rank ") and " of " in l.split(":")[0]]
        assert len(lines) == world and all("copies into pages rank 0 made" in l for l in lines), lines
        assert b"fxh timing rank 0: the job's pages:" in res[0][2]
    # irregular input in one rank's range: EVERY rank leaves (flag in the exchanged block), rank 0 runs the input as one stream -> the reference's behaviour
    k0 = text.index(b"\n@", int(len(text) * 0.7)) + 1
    bad = text[:k0] + b"#" + text[k0 + 1:]
    inp.write_bytes(bad)
    argv = ["fastq_quality_trim_filter", "-t", "20", "-l", "30", "-q", "20", "-p", "80", "-v"]
    w = _run([os.path.join(tools, argv[0])] + argv[1:] + ["-i", str(inp), "-o", str(tmp_path / "bad_single.fq")], b"", extra_env={"FXH_ONE_FILE": "0"})
    res = _rank_job(tools, argv, inp, tmp_path / "bad_ranks.fq", world, tmp_path)
    assert w[0] == 1 and res[0][0] == 1 and all(rc == 0 for rc, _, _ in res[1:]), [e[-300:] for _, _, e in res]
    assert _msg(res[0][2].splitlines()[-1]) == _msg(w[2].splitlines()[-1])
    assert (tmp_path / "bad_ranks.fq").read_bytes() == (tmp_path / "bad_single.fq").read_bytes()
    res = _rank_job(tools, argv, inp, tmp_path / "bad_paged.fq", world, tmp_path, extra={"FXH_ONE_FILE_SINK": "map", "FXH_ONE_FILE_WINDOW_MB": "1"})      # (rank 0's allocator is stopped, its pages go)
    assert res[0][0] == 1 and all(rc == 0 for rc, _, _ in res[1:]) and (tmp_path / "bad_paged.fq").read_bytes() == (tmp_path / "bad_single.fq").read_bytes()
    # the clipper on reads that are not all of one length: the same way out, and the one-stream run goes serial where it must
    lines = clip_text.split(b"\n")[:-1]
    for i in range(len(lines) // 4 * 6 // 10 * 4, len(lines), 8):
        lines[i + 1] = lines[i + 1][:61]; lines[i + 3] = lines[i + 3][:61]
    data = b"\n".join(lines) + b"\n"
    inp.write_bytes(data)
    cargv = ["fastx_clipper", "-a", "AGATCGGAAGAGC", "-l", "15", "-v"]
    res = _rank_job(tools, cargv, inp, tmp_path / "clip_ranks.fq", world, tmp_path)
    assert all(rc == 0 for rc, _, _ in res), [e[-300:] for _, _, e in res]
    if REF:
        ref = _run([REF] + cargv, data)
        assert res[0][1] == ref[2] and (tmp_path / "clip_ranks.fq").read_bytes() == ref[1]
    # an input no rank mode takes (a pipe): rank 0 does the job, the others step aside without touching the file
    env = dict(os.environ, LD_LIBRARY_PATH=STUB_DIR, FXH_THREADS="2", FXH_WORLD=str(world))
    outp = tmp_path / "pipe_ranks.fq"
    ps = [subprocess.Popen([os.path.join(tools, argv[0])] + argv[1:] + ["-o", str(outp)], stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(env, FXH_RANK=str(r))) for r in range(world)]
    outs = [p.communicate(text, timeout=120) for p in ps]
    assert all(p.returncode == 0 for p in ps) and outp.read_bytes() == (tmp_path / "single0").read_bytes() and all(o == b"" for o, _ in outs[1:])


def test_rank_job_whose_input_one_rank_cannot_cut_starts_in_no_rank(tools, tmp_path):
    """Whether a job runs by ranks is decided before the ranks meet, so it must be the same decision everywhere: a region that cannot be cut into chunks
    (here: records far longer than a chunk) lies in rank 1's byte range ONLY.  Every rank looks at every rank's range, so rank 0 sees it too: no rank goes
    to the rendezvous (a rank that did would wait there for the one that left), ranks 1 and 2 leave with 0 and rank 0 runs the input as one stream --
    the single-process bytes and report.  Bounded by the test's own time-out: a hang is a failure."""
    import numpy as np
    rng = np.random.default_rng(5)
    def rec(i, n):
        b = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n))
        return b"@r%d\n" % i + b + b"\n+\n" + b"I" * n + b"\n"
    parts = [rec(i, 100) for i in range(3000)]
    text = b"".join(parts[:1500]) + b"".join(rec(10000 + i, 20000) for i in range(6)) + b"".join(parts[1500:])      # the long records sit in the middle third
    inp = tmp_path / "in.fq"
    inp.write_bytes(text)
    argv = ["fastx_reverse_complement", "-v"]
    one = _run([os.path.join(tools, argv[0])] + argv[1:] + ["-i", str(inp), "-o", str(tmp_path / "one.fq")], b"", buf_mb="1", extra_env={"FXH_ONE_FILE": "0"})
    assert one[0] == 0
    res = _rank_job(tools, argv, inp, tmp_path / "job.fq", 3, tmp_path, extra={"FXH_STRAND_KB": "4", "FXH_RENDEZVOUS_TIMEOUT": "20", "FXH_RANK_TIMEOUT": "20"})
    assert [rc for rc, _, _ in res] == [0, 0, 0], [e[-300:] for _, _, e in res]
    assert (tmp_path / "job.fq").read_bytes() == (tmp_path / "one.fq").read_bytes() and res[0][1] == one[1]
    assert res[1][1] == b"" and res[2][1] == b""


def test_rank_job_is_done_when_every_rank_has_written(tools, tmp_path):
    """Rank 0's exit code is the job's.  After the ranks have written their parts they meet a second time (each rank's errno, 0 = written): a rank that could not
    write -- here: a file size limit on rank 1 alone -- says so there, rank 0 names it and leaves with 1; a rank that DIED on the way never arrives, and
    what ends rank 0's wait is the transport's own failure or, where the transport waits for ever (RCCL does), the watch (FXH_RANK_TIMEOUT)."""
    import resource
    import signal
    import emu_py
    fake = emu_py.build_fake_rccl()
    text = fo.synth_fastq(47, 0, 40000, 100, False)
    inp = tmp_path / "in.fq"
    inp.write_bytes(text)
    argv = ["fastx_reverse_complement", "-v"]

    def job(out, limit_rank1, ignore_xfsz, extra):
        procs = []
        for r in range(2):
            env = dict(os.environ, LD_LIBRARY_PATH=STUB_DIR + os.pathsep + fake, FXH_THREADS="2", FXH_RANK=str(r), FXH_WORLD="2", FXH_STRAND_KB="256", FXH_STRANDS="2",
                       FXH_DRAIN_MB="1", FXG_EMU_DEVICES="2", **extra)

            def pre(r=r):
                if r == 1 and limit_rank1:
                    if ignore_xfsz:
                        signal.signal(signal.SIGXFSZ, signal.SIG_IGN)       # (an ignored signal stays ignored across exec: the write then fails with EFBIG)
                    resource.setrlimit(resource.RLIMIT_FSIZE, (65536, 65536))
            procs.append(subprocess.Popen([os.path.join(tools, argv[0])] + argv[1:] + ["-i", str(inp), "-o", str(out)], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                          env=env, preexec_fn=pre))
        return [(p.returncode, o, e) for p in procs for o, e in [p.communicate(timeout=120)]]

    # the undisturbed job, for reference
    ok = job(tmp_path / "ok.fq", False, False, {})
    assert [rc for rc, _, _ in ok] == [0, 0] and len((tmp_path / "ok.fq").read_bytes()) == len(text)
    # rank 1 cannot write (EFBIG): both ranks learn it in the second exchange; rank 0 names the rank and the reason, prints NO report, exit code 1
    res = job(tmp_path / "efbig.fq", True, True, {})
    assert res[0][0] == 1 and res[1][0] == 1, [e[-300:] for _, _, e in res]
    assert b"rank 1 of 2 could not write its part of the output (File too large)" in res[0][2] and b"is incomplete" in res[0][2] and res[0][1] == b""
    assert b"rank 1 of 2: writing output failed: File too large" in res[1][2]
    # rank 1 is killed on the way (SIGXFSZ): it never joins the second exchange.  The fake transport gives up after 2 s ...
    res = job(tmp_path / "dead.fq", True, False, {"FXG_FAKE_RCCL_TIMEOUT_S": "2"})
    assert res[1][0] == -signal.SIGXFSZ and res[0][0] == 1 and res[0][1] == b"", [e[-300:] for _, _, e in res]
    assert b"waiting for every rank to have written its part failed" in res[0][2]
    # ... and with a transport that waits for ever, the watch ends rank 0
    res = job(tmp_path / "dead2.fq", True, False, {"FXG_FAKE_RCCL_TIMEOUT_S": "0", "FXH_RANK_TIMEOUT": "2"})
    assert res[1][0] == -signal.SIGXFSZ and res[0][0] == 1 and res[0][1] == b"", [e[-300:] for _, _, e in res]
    assert b"no answer from the other ranks within 2 s (waiting for every rank to have written its part)" in res[0][2]


def test_pipes_as_the_references_users_run_them(tools, tmp_path):
    """`trimmer | filter` and `cat in | tool > out` (the Galaxy wrappers' form): both ends of a pipe are raised to the system's limit (F_SETPIPE_SZ); a block goes
    into the pipe as pieces that writer threads write() into private pipes side by side and that are moved on into the output pipe in order by splice() -- page
    references, no copy, and the kernel's own pages, so any reader is safe --; the reading side deals the incoming pages out to private pipes the same way and
    copies them out by several threads.  Same bytes with the default 64 KB pipe, without the private pipes, with many of them, with tiny and with large blocks; a
    splicing middleman in between; a reader that stops early (head) ends the writer like any tool."""
    text = fo.synth_fastq(47, 0, 60000, 100, False)
    inp = tmp_path / "in.fq"
    inp.write_bytes(text)
    t1 = [os.path.join(tools, "fastq_quality_trimmer"), "-t", "20", "-l", "30"]
    t2 = [os.path.join(tools, "fastq_quality_filter"), "-q", "20", "-p", "80", "-v"]
    ref = _run(t2, _run(t1, text)[1])
    assert ref[0] == 0 and len(ref[1]) > 1_000_000
    base = dict(os.environ, LD_LIBRARY_PATH=STUB_DIR, FXH_THREADS="2")
    for extra in ({}, {"FXH_NO_PIPE_TUNING": "1"}, {"FXH_PIPE_WRITERS": "2", "FXH_PIPE_MB": "1", "FXH_READ_BUFFER_MB": "4"}, {"FXH_READ_BUFFER_MB": "1"}, {"FXH_READ_BUFFER_MB": "1", "FXH_NO_PIPE_TUNING": "1"}, {"FXH_HOST_PARSE": "1"},
                  {"FXH_NO_PIPE_FANOUT": "1"}, {"FXH_PIPE_READERS": "8", "FXH_PIPE_WRITERS": "4", "FXH_PIPE_MB": "1", "FXH_READ_BUFFER_MB": "4"}):
        env = dict(base, **extra)
        out = tmp_path / "piped.fq"
        p1 = subprocess.Popen(t1 + ["-i", str(inp)], stdout=subprocess.PIPE, env=env)
        p2 = subprocess.Popen(t2 + ["-o", str(out)], stdin=p1.stdout, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        p1.stdout.close()
        o, e = p2.communicate(timeout=120)
        assert p1.wait() == 0 and p2.returncode == 0, e[-300:]
        assert out.read_bytes() == ref[1] and o == ref[2], extra              # (-v goes to stdout when -o names a file)
        # cat in | tool > out
        with open(inp, "rb") as f, open(tmp_path / "redir.fq", "wb") as g:
            c = subprocess.Popen(["cat"], stdin=f, stdout=subprocess.PIPE)
            p = subprocess.Popen(t1, stdin=c.stdout, stdout=g, env=env)
            c.stdout.close()
            assert p.wait(timeout=120) == 0 and c.wait() == 0
        assert (tmp_path / "redir.fq").read_bytes() == _run(t1, text)[1]
    # a splicing middleman (what pv does): the pages in the writer's pipe travel on into a second pipe and stay there while the reader dawdles.  They are the
    # kernel's own pipe pages (the writer's threads copied into them) -- never pages of a buffer the tool writes to again -- so the bytes are right
    fwd = tmp_path / "fwd.py"
    fwd.write_text("import os, time, fcntl\nfcntl.fcntl(1, 1031, 1 << 20)\nn = 0\nwhile True:\n    k = os.splice(0, 1, 1 << 20)\n    if k == 0: break\n    n += k\n    if (n >> 20) % 4 == 0: time.sleep(0.002)\n")
    env = dict(base, FXH_READ_BUFFER_MB="4", FXH_PIPE_MB="1")
    p1 = subprocess.Popen(t1 + ["-i", str(inp)], stdout=subprocess.PIPE, env=env)
    pm = subprocess.Popen([sys.executable, str(fwd)], stdin=p1.stdout, stdout=subprocess.PIPE)
    p1.stdout.close()
    got = bytearray()
    while True:
        b = pm.stdout.read(1 << 16)
        if not b:
            break
        got += b
        time.sleep(0.0002)
    assert p1.wait() == 0 and pm.wait() == 0 and bytes(got) == _run(t1, text)[1]
    # the downstream side stops reading: the tool dies of SIGPIPE (or reports the failed write), it does not hang
    p1 = subprocess.Popen(t1 + ["-i", str(inp)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=base)
    h = subprocess.Popen(["head", "-c", "100000"], stdin=p1.stdout, stdout=subprocess.PIPE)
    p1.stdout.close()
    got = h.communicate(timeout=60)[0]
    assert got == _run(t1, text)[1][:100000] and p1.wait(timeout=60) != 0


@pytest.mark.skipif(REF is None, reason="oracle/_ref/fxref not built")
def test_clipper_debug_dump(tools, tmp_path):
    """fastx_clipper -D / -D -D (fastx_clipper.cpp:270-275): every read's SequenceAlignmentResults (sequence_alignment.cpp:15-84) and, given twice, the whole
    score / origin / match matrix (:169-230) on stdout -- the alignment strings, the float formatting of std::cout (default, then `fixed` with one digit for
    good once the first matrix cell has been printed), the never-shrinking matrix and the stale query tail on ragged input, adapters with N -- byte for byte
    against the real libfastx; the records and the -v report as without -D."""
    rng = np.random.default_rng(5)
    lines = fo.synth_fastq(3, 0, 400, 60, True).split(b"\n")[:-1]
    for i in range(0, len(lines), 4):
        if (i // 4) % 3 == 1:
            L = int(rng.integers(5, 60)); lines[i + 1] = lines[i + 1][:L]; lines[i + 3] = lines[i + 3][:L]
    data = b"\n".join(lines) + b"\n"
    fasta = b"".join(b">%d-%d\n%s\n" % (i, 1 + i % 3, lines[4 * i + 1]) for i in range(60))
    n = 0
    for text in (data, fasta):
        for flags in (["-D"], ["-D", "-D"], ["-D", "-c", "-v"], ["-D", "-D", "-n", "-k"], ["-D", "-M", "6", "-d", "2", "-v"]):
            for ad in ("AGATCGGAAGAGC", "ANNTCGNA", "TGGAATTCTCGGGTGCCAAGGAACTCCAGTCAC"):
                argv = ["fastx_clipper", "-a", ad, "-l", "5"] + flags
                w = _run([REF] + argv + ["-o", str(tmp_path / "ref.out")], text)
                g = _run([os.path.join(tools, argv[0])] + argv[1:] + ["-o", str(tmp_path / "my.out")], text)
                assert (g[0], g[1]) == (w[0], w[1]), (flags, ad, len(g[1]), len(w[1]))
                assert (tmp_path / "my.out").read_bytes() == (tmp_path / "ref.out").read_bytes()
                n += 1
    assert n == 30
    # without -D the same command lines are the engine's: same records, same report, nothing else on stdout
    g = _run([os.path.join(tools, "fastx_clipper"), "-a", "AGATCGGAAGAGC", "-l", "5", "-v", "-o", str(tmp_path / "gpu.out")], data)
    w = _run([REF, "fastx_clipper", "-a", "AGATCGGAAGAGC", "-l", "5", "-v", "-o", str(tmp_path / "ref.out")], data)
    assert (g[0], g[1]) == (w[0], w[1]) and (tmp_path / "gpu.out").read_bytes() == (tmp_path / "ref.out").read_bytes()
