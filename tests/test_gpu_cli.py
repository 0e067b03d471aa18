"""GPU tier (-m gpu): the five command-line tools (C host layer + HIP engine) against the reference.

* every Galaxy known-answer pair of the hot tools, byte for byte;
* the seeded synthetic cases of tests/golden/cases.json by md5 (reference output);
* where oracle/_ref/fxref travelled to the box: fuzzed text incl. -v reports, exit codes and the
  "flush everything before the bad record, then fail with the reference's message" behaviour.
"""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from helpers import GOLDEN, md5
from oracle import fxoracle_py as fo

pytestmark = pytest.mark.gpu
HOST = os.path.join(ROOT, "fastx_toolkit_amd", "host")
REF = fo.ref_binary()


@pytest.fixture(scope="module")
def tools():
    subprocess.check_call(["make", "-s", "-C", HOST])
    return os.path.join(HOST, "bin")


PARSE_ENV = {}          # set per test run by the `parse_mode` fixture: {} = device-side parse/format, FXH_HOST_PARSE=1 = host parser


def _run(cmd, data, env=None):
    env = dict(env or os.environ, **PARSE_ENV)
    p = subprocess.run(cmd, input=data, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=180)   # a hang must fail fast
    return p.returncode, p.stdout, p.stderr


@pytest.fixture(scope="module", params=["device-parse", "host-parse"], autouse=True)
def parse_mode(request):
    PARSE_ENV.clear()
    if request.param == "host-parse":
        PARSE_ENV["FXH_HOST_PARSE"] = "1"
    yield request.param
    PARSE_ENV.clear()


def _msg(err):
    lines = [l for l in err.split(b"\n") if b"amdgpu.ids" not in l and not l.startswith(b"fxh timing")]
    return b"\n".join(lines).split(b": ", 1)[-1]


def test_galaxy_known_answers(tools, cases):
    for g in cases["galaxy"]:
        inp = open(os.path.join(GOLDEN, "galaxy", g["input"]), "rb").read()
        exp = open(os.path.join(GOLDEN, "galaxy", g["expect"]), "rb").read()
        rc, out, err = _run([os.path.join(tools, g["cmd"][0])] + g["cmd"][1:], inp)
        assert rc == 0, err
        assert out == exp, g["name"]


def test_synthetic_cases_md5(tools, cases):
    for c in cases["synthetic"]:
        if c["n"] > 200000:
            continue
        text = fo.synth_fastq(c["seed"], 0, c["n"], c["L"], c["adapter"])
        for cmd in c["chain"]:
            rc, text, err = _run([os.path.join(tools, cmd[0])] + cmd[1:], text)
            assert rc == 0, err
        assert md5(text) == c["output_md5"], c["name"]
    for c in cases["varlen"]:
        text = open(os.path.join(GOLDEN, "synthetic", c["name"] + ".fq"), "rb").read()
        for cmd in c["chain"]:
            rc, text, err = _run([os.path.join(tools, cmd[0])] + cmd[1:], text)
            assert rc == 0, err
        assert text == open(os.path.join(GOLDEN, "synthetic", c["name"] + ".out"), "rb").read(), c["name"]


def test_files_reports_gzip_and_small_buffers(tools, tmp_path):
    text = fo.synth_fastq(31, 0, 30000, 100, True)
    inp, outp = tmp_path / "in.fq", tmp_path / "out.fq"
    inp.write_bytes(text)
    env = dict(os.environ, FXH_READ_BUFFER_MB="1")             # forces many engine calls with records straddling refills
    rc, out, err = _run([os.path.join(tools, "fastq_quality_trimmer"), "-t", "25", "-l", "40", "-v", "-i", str(inp), "-o", str(outp)], b"", env)
    assert rc == 0, err
    exp, r = None, None
    from helpers import oracle_params, text_through
    exp, r = text_through(fo.run_pipeline, text, oracle_params(dict(stages=2, qt_threshold=25, qt_min_len=40)))
    assert outp.read_bytes() == exp
    kept = int(r["counters"][1])
    assert out.decode() == ("Minimum Quality Threshold: 25\nMinimum Length: 40\nInput: 30000 reads.\nOutput: %d reads.\n"
                            "discarded %d (%d%%) too-short reads.\n" % (kept, 30000 - kept, (30000 - kept) * 100 // 30000))   # -o => report on stdout
    rc, out, err = _run([os.path.join(tools, "fastx_reverse_complement"), "-z", "-i", str(inp)], b"")
    assert rc == 0
    import gzip
    exp, _ = text_through(fo.run_pipeline, text, oracle_params(dict(stages=8)))
    assert gzip.decompress(out) == exp


@pytest.mark.skipif(REF is None, reason="oracle/_ref/fxref not on this box")
def test_fuzz_cli_vs_reference(tools):
    rng = np.random.default_rng(9)
    ad = b"AGATCGGAAGAGC"
    for trial in range(24):
        L = Lmax = int(rng.integers(20, 90))
        recs = []
        for i in range(int(rng.integers(50, 400))):
            if trial % 3 == 1:
                L = int(rng.integers(8, Lmax + 1))   # ragged input: the clipper then depends on the reads before (N3)
            s = rng.choice(np.frombuffer(b"ACGTN", np.uint8), size=L, p=[.24, .24, .24, .24, .04])
            if rng.random() < 0.5:
                pos = int(rng.integers(0, L + 1)); k = min(len(ad), L - pos)
                s[pos:pos + k] = np.frombuffer(ad, np.uint8)[:k]
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
            env = dict(os.environ, FXH_THREADS=str([16, 1, 3, 7][trial % 4]), FXH_READ_BUFFER_MB="1")
            rc, out, err = _run([os.path.join(tools, argv[0])] + argv[1:], data, env)
            rrc, rout, rerr = _run([REF] + argv, data)
            assert (rc, out) == (rrc, rout), (trial, argv)
            assert _msg(err) == _msg(rerr), (trial, argv)


def test_reader_rules_through_the_batch_path(tools):
    """R1-R9 on the GPU build: CRLF, no final newline, numeric qualities, FASTA with collapsed ids -- every tool against the reference driver."""
    from test_host_cli_emulated import _odd_inputs
    rng = np.random.default_rng(78)
    ad = "AGATCGGAAGAGC"
    for kind, data in _odd_inputs(rng).items():
        fasta = kind.startswith("fasta")
        argvs = [["fastx_trimmer", "-f", "3", "-l", "20"], ["fastx_reverse_complement"], ["fastx_clipper", "-a", ad, "-l", "5", "-n", "-v"], ["fastx_artifacts_filter", "-v"]]
        if not fasta:
            argvs += [["fastq_quality_trimmer", "-t", "18", "-l", "8", "-v"], ["fastq_quality_filter", "-q", "15", "-p", "60", "-v"],
                      ["fastq_masker", "-q", "12"], ["fastq_to_fasta", "-r"], ["fastx_quality_stats"]]
        for argv in argvs:
            rc, out, err = _run([os.path.join(tools, argv[0])] + argv[1:], data, dict(os.environ, FXH_THREADS="5", FXH_READ_BUFFER_MB="1", FXH_TIMING="1"))
            rrc, rout, rerr = _run([REF] + argv, data)
            assert (rc, out) == (rrc, rout), (kind, argv, err[-200:], rerr[-200:])
            assert _msg(err) == _msg(rerr), (kind, argv)
            if not PARSE_ENV and argv[0] != "fastx_quality_stats" and argv[:2] != ["fastq_to_fasta", "-r"]:
                # CRLF, numeric qualities and FASTA are handled on the device: no block may have fallen back to the host parser
                assert b"device parse" in err and b" 0 host-parsed blocks" in err, (kind, argv, err[-300:])


def test_minimal_c_caller_of_the_abi(tools):
    """host/examples/abi_minimal.c: plain C against include/fxg.h -- its counters are the oracle's."""
    from helpers import oracle_params
    n, L = 200000, 150
    rc, out, err = _run([os.path.join(tools, "abi_minimal"), str(n), str(L)], b"")
    assert rc == 0, err
    b, q = fo.synth_batch(2, 0, n, L)
    c = fo.run_pipeline(b, q, None, oracle_params(dict(stages=6, qt_threshold=20, qt_min_len=30, qf_min_quality=20, qf_min_percent=80)))["counters"]
    assert out.decode().split() == ["input", str(n), "kept", str(int(c[1])), "kept_bases", str(int(c[2])), "qtrim_dropped", str(int(c[8])),
                                   "qfilter_dropped", str(int(c[9]))]


def test_quality_stats_galaxy_known_answer(tools):
    inp = open(os.path.join(GOLDEN, "galaxy", "fastq_stats1.fastq"), "rb").read()
    exp = open(os.path.join(GOLDEN, "galaxy", "fastq_stats1.out"), "rb").read()
    rc, out, err = _run([os.path.join(tools, "fastx_quality_stats"), "-Q", "64"], inp)
    assert (rc, out) == (0, exp), err


def test_lanes_devices_and_the_fused_tool(tools, tmp_path):
    """Multi-GPU front end of the tools: blocks dealt round-robin to lanes (one context each) over FXG_DEVICES and collected in input
    order.  Two contexts on this box's one GPU stand in for two GPUs (the code path is the same; the driver's 8-GPU node is where
    they would be different devices).  Output and -v report must be byte-identical to the single-lane run whatever the lane count
    and block size; a damaged record fails like the reference after everything before it was written.  Also: the one-pass
    fastq_quality_trim_filter writes what the trimmer | filter pipe writes."""
    text = fo.synth_fastq(2, 0, 120000, 150, False)                      # ~38 MB
    argv = ["fastq_quality_trimmer", "-t", "20", "-l", "30", "-v"]
    base = _run([os.path.join(tools, argv[0])] + argv[1:], text, dict(os.environ, FXH_LANES="1"))
    assert base[0] == 0
    for env in ({"FXH_LANES": "3"}, {"FXG_DEVICES": "0,0", "FXH_LANES": "2"}, {"FXG_DEVICES": "all"}):
        for buf in ("1", "8"):
            got = _run([os.path.join(tools, argv[0])] + argv[1:], text, dict(os.environ, FXH_READ_BUFFER_MB=buf, **env))
            assert (got[0], got[1], _msg(got[2])) == (base[0], base[1], _msg(base[2])), (env, buf)
    k = text.index(b"\n@", 20_000_000) + 1
    bad = text[:k] + b"#" + text[k + 1:]
    want = _run([os.path.join(tools, argv[0])] + argv[1:], bad, dict(os.environ, FXH_LANES="1"))
    assert want[0] == 1 and len(want[1]) > 5_000_000
    if REF:
        ref = _run([REF] + argv, bad)
        assert (want[0], want[1]) == (ref[0], ref[1]) and _msg(want[2]) == _msg(ref[2])
    got = _run([os.path.join(tools, argv[0])] + argv[1:], bad, dict(os.environ, FXG_DEVICES="0,0", FXH_LANES="2", FXH_READ_BUFFER_MB="2"))
    assert (got[0], got[1]) == (want[0], want[1]) and _msg(got[2]) == _msg(want[2])
    # fused one-pass tool == the shell pipe, output and reports
    t = _run([os.path.join(tools, "fastq_quality_trimmer"), "-t", "20", "-l", "30", "-v"], text)
    f = _run([os.path.join(tools, "fastq_quality_filter"), "-q", "20", "-p", "80", "-v"], t[1])
    one = _run([os.path.join(tools, "fastq_quality_trim_filter"), "-t", "20", "-l", "30", "-q", "20", "-p", "80", "-v"], text)
    assert one[0] == 0 and one[1] == f[1]
    clean = lambda e: b"\n".join(l for l in e.split(b"\n") if b"amdgpu.ids" not in l)
    assert clean(one[2]) == clean(t[2]) + clean(f[2])


def test_sharded_run_and_the_three_stage_tool_on_the_gpu_build(tools, tmp_path):
    """FXH_PARTS on the real engine: parts concatenate to the one-stream output, and fastx_clip_trim_filter (config 5 in one pass) ==
    the three-tool pipe of the real libfastx."""
    text = fo.synth_fastq(5, 0, 150000, 150, True)                       # ~48 MB
    inp = tmp_path / "in.fq"
    inp.write_bytes(text)
    ad = "AGATCGGAAGAGC"
    for argv, penv in ((["fastq_quality_trim_filter", "-t", "20", "-l", "30", "-q", "20", "-p", "80", "-v"], {}),
                       (["fastx_clip_trim_filter", "-a", ad, "-l", "15", "-n", "-t", "20", "-m", "30", "-q", "20", "-p", "80", "-v"], {}),       # fixed-length input: parallel by itself (round 4)
                       (["fastx_reverse_complement", "-v"], {})):
        single = tmp_path / "single.fq"
        want = _run([os.path.join(tools, argv[0])] + argv[1:] + ["-i", str(inp), "-o", str(single)], b"", dict(os.environ, **penv))
        assert want[0] == 0
        for k in (2, 4):
            pat = str(tmp_path / ("p%d.%%r.fq" % k))
            got = _run([os.path.join(tools, argv[0])] + argv[1:] + ["-i", str(inp), "-o", pat], b"", dict(os.environ, FXH_PARTS=str(k), FXH_READ_BUFFER_MB="4", FXH_TIMING="1", **penv))
            assert got[0] == 0 and got[1] == want[1], (argv[0], k, got[2][-300:])
            if not PARSE_ENV:
                assert got[2].count(b"fxh timing part") == k, got[2][-400:]
            assert b"".join(open(pat.replace("%r", str(r)), "rb").read() for r in range(k)) == single.read_bytes(), (argv[0], k)
    # (what a sharded run does with damaged input -- abandon the attempt, run as one stream: the last test of this file, on the real runtime)
    if REF:                                                               # config 5 as the reference runs it: three processes in a pipe
        small = text[:3_000_000]
        small = small[:small.rindex(b"\n@") + 1]
        rc_ = _run([REF, "fastx_clipper", "-a", ad, "-l", "15", "-n", "-v"], small)
        rt = _run([REF, "fastq_quality_trimmer", "-t", "20", "-l", "30", "-v"], rc_[1])
        rf = _run([REF, "fastq_quality_filter", "-q", "20", "-p", "80", "-v"], rt[1])
        one = _run([os.path.join(tools, "fastx_clip_trim_filter"), "-a", ad, "-l", "15", "-n", "-t", "20", "-m", "30", "-q", "20", "-p", "80", "-v"], small)
        clean = lambda e: b"\n".join(l for l in e.split(b"\n") if b"amdgpu.ids" not in l)
        assert one[0] == 0 and one[1] == rf[1] and clean(one[2]) == rc_[2] + rt[2] + rf[2]


def test_flag_errors_and_usage_on_the_gpu_build(tools):
    """F1-F6 against the real libfxg.so (the CPU tier runs the same checks against the emulation stub): exit codes, usage text, and the
    reference's messages where fxref is on the box."""
    from helpers import FLAG_ERROR_CASES as cases
    for argv, data in cases:
        rc, out, err = _run([os.path.join(tools, argv[0])] + argv[1:], data)
        assert rc == 1, argv
        if REF:
            rrc, rout, rerr = _run([REF] + argv, data)
            assert (rc, out) == (rrc, rout), argv
            assert _msg(err) == _msg(rerr), argv
    rc, out, _ = _run([os.path.join(tools, "fastx_clipper"), "-h"], b"")
    assert rc == 1 and out.startswith(b"usage: fastx_clipper")                                     # -h exits 1 (F5)
    if REF:                                                                                         # -D: the reference's dump, from the host (host/fxh_clip_debug.c)
        assert _run([os.path.join(tools, "fastx_clipper"), "-a", "ACGT", "-D"], b"@r\nA\n+\nI\n")[:2] == _run([REF, "fastx_clipper", "-a", "ACGT", "-D"], b"@r\nA\n+\nI\n")[:2]
    rc, out, err = _run([os.path.join(tools, "fastq_quality_filter"), "-q", "94", "-v"], b"@r\nACGT\n+\nIIII\n")   # F2: -p omitted, -q > 93 drops everything
    assert rc == 0 and out == b""
    rc, out, err = _run([os.path.join(tools, "fastq_quality_filter"), "-q", "40"], b"@r\nACGT\n+\n!!!!\n")         # F2: -p omitted, everything passes
    assert rc == 0 and out == b"@r\nACGT\n+\n!!!!\n"


def test_output_longer_than_input_and_ragged_blocks(tools):
    """An empty third line still gets its '+' on output (fastx.c:460): dense 7-byte records grow by one byte each, so the formatted
    block is longer than the text it came from; and one long read among short ones must not make the rows explode."""
    data = b"@\nA\n\nI\n" * 300000
    rc, out, err = _run([os.path.join(tools, "fastq_quality_trimmer"), "-t", "20", "-l", "1"], data, dict(os.environ, FXH_READ_BUFFER_MB="1"))
    assert rc == 0 and out == b"@\nA\n+\nI\n" * 300000
    rng = np.random.default_rng(6)
    recs = [b"@s%d\nACGTACGTAC\n+\nIIIIIIIIII\n" % i for i in range(100000)]
    longr = b"@long\n" + rng.choice(np.frombuffer(b"ACGT", np.uint8), size=20000).tobytes() + b"\n+\n" + b"I" * 20000 + b"\n"
    data = b"".join(recs[:50000]) + longr + b"".join(recs[50000:])
    rc, out, err = _run([os.path.join(tools, "fastx_reverse_complement")], data)
    assert rc == 0 and out.count(b"\n") == 4 * 100001
    if REF:
        assert out == _run([REF, "fastx_reverse_complement"], data)[1]


@pytest.mark.skipif(REF is None, reason="oracle/_ref/fxref not built")
def test_clipper_long_fixed_length_reads_on_the_gpu_build(tools):
    """fastx_clipper / fastx_clip_trim_filter on reads of ONE length beyond 160 bases: the lanes pack the rows with a stride rounded up to a
    multiple of four (host/fxh_lanes.c) so that the clip kernel may read them where they are (csrc/fxg_plan.h: clip_global).  Lengths that are
    and are not multiples of four, with and without -n, against the real libfastx."""
    ad = "AGATCGGAAGAGC"
    for L in (161, 250, 251, 300, 301):
        data = fo.synth_fastq(70 + L, 0, 3000, L, True)
        for argv in (["fastx_clipper", "-a", ad, "-l", "15", "-v"], ["fastx_clipper", "-a", ad, "-l", "15", "-n", "-v"]):
            ref = _run([REF] + argv, data)
            got = _run([os.path.join(tools, argv[0])] + argv[1:], data, dict(os.environ, FXH_TIMING="1"))
            assert (got[0], got[1]) == (0, ref[1]), (L, argv, got[2][-300:])
            assert b" 0 host-parsed blocks" in got[2], got[2][-300:]
        chain = [["fastx_clipper", "-a", ad, "-l", "15", "-n"], ["fastq_quality_trimmer", "-t", "20", "-l", "30"], ["fastq_quality_filter", "-q", "20", "-p", "80"]]
        want = data
        for c in chain:
            want = _run([REF] + c, want)[1]
        got = _run([os.path.join(tools, "fastx_clip_trim_filter"), "-a", ad, "-l", "15", "-n", "-t", "20", "-m", "30", "-q", "20", "-p", "80"], data)
        assert (got[0], got[1]) == (0, want), (L, got[2][-300:])


@pytest.mark.skipif(REF is None, reason="oracle/_ref/fxref not built")
def test_clipper_default_command_line_on_ragged_input(tools, tmp_path):
    """fastx_clipper with no environment variable on the real engine: parallel lanes while the reads have one length, the reference's one
    aligner (seeded with the last record before) from the first block that differs -- byte-identical to the real libfastx clipper when
    the ragged reads start at 0 %, 50 %, 99 % of the input or never."""
    rng = np.random.default_rng(78)
    ad = "AGATCGGAAGAGC"
    lines = fo.synth_fastq(62, 0, 60000, 150, True).split(b"\n")
    recs = [b"\n".join(lines[4 * i:4 * i + 4]) + b"\n" for i in range(60000)]

    def ragged(rec):
        L = int(rng.integers(20, 150))
        l = rec.split(b"\n")
        return b"\n".join([l[0], l[1][:L], l[2], l[3][:L]]) + b"\n"
    for name, start in (("never", None), ("at_0", 0.0), ("at_50", 0.5), ("at_99", 0.99)):
        k = len(recs) if start is None else int(len(recs) * start)
        data = b"".join(recs[:k]) + b"".join(ragged(r) if rng.random() < 0.5 else r for r in recs[k:])
        ref = _run([REF, "fastx_clipper", "-a", ad, "-l", "15", "-v"], data)
        for env in ({"FXH_READ_BUFFER_MB": "2"}, {"FXH_READ_BUFFER_MB": "2", "FXH_LANES": "3"}):
            got = _run([os.path.join(tools, "fastx_clipper"), "-a", ad, "-l", "15", "-v"], data, dict(os.environ, FXH_TIMING="1", **env))
            assert (got[0], got[1]) == (0, ref[1]), (name, env, got[2][-400:])
            if not PARSE_ENV:
                assert (b"one aligner with history from there on" in got[2]) == (name != "never"), (name, got[2][-400:])


@pytest.mark.skipif(REF is None, reason="oracle/_ref/fxref not built")
def test_sharded_run_abandoned_on_the_real_runtime(tools, tmp_path):
    """LAST in this file on purpose.  FXH_PARTS=4 over input the device path does not take: the sharded attempt (a forked child with four
    parts, each with its own contexts, streams and page-locked buffers on the real HIP runtime) is abandoned -- every thread joined, every
    context destroyed, the parts emptied, _exit(99) -- and the parent, which has not touched the GPU, runs the input as one stream:
    exit code, message, -v report and partial output are the reference's.  (The first version of this, in round 3, re-exec'd the
    process with device work in flight and took two GPU boxes down; it has only ever run against the emulation stub since.)"""
    if PARSE_ENV:
        pytest.skip("the sharded run is the device-parse path")
    text = fo.synth_fastq(47, 0, 1_200_000, 100, False)                  # ~280 MB, 1.2 M reads
    k0 = text.index(b"\n@", int(len(text) * 0.4)) + 1                    # inside part 1 of 0..3 ("part 2 of 4")
    cases = {"damaged_record": text[:k0] + b"#" + text[k0 + 1:], "ragged_end": text[:-150],
             "ragged_clipper_input": None}
    argv = ["fastq_quality_trim_filter", "-t", "20", "-l", "30", "-q", "20", "-p", "80", "-v"]
    inp = tmp_path / "in.fq"
    for name in ("damaged_record", "ragged_end"):
        inp.write_bytes(cases[name])
        ref1 = _run([REF, "fastq_quality_trimmer", "-t", "20", "-l", "30"], cases[name])
        ref2 = _run([REF, "fastq_quality_filter", "-q", "20", "-p", "80"], ref1[1])          # what the reference leaves behind before it fails
        pat = str(tmp_path / (name + ".%r.fq"))
        p = subprocess.run([os.path.join(tools, argv[0])] + argv[1:] + ["-i", str(inp), "-o", pat], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           env=dict(os.environ, FXH_PARTS="4", FXH_TIMING="1"), timeout=120)
        assert p.returncode == 1 == ref1[0], p.stderr[-600:]
        assert b"fxh timing parts: abandoned, contexts destroyed, parts emptied" in p.stderr
        assert _msg(p.stderr.split(b"fxh timing parts: abandoned, contexts destroyed, parts emptied\n")[-1]) == _msg(ref1[2]), (p.stderr[-400:], ref1[2])
        out = b"".join(open(pat.replace("%r", str(r)), "rb").read() for r in range(4))
        assert out == ref2[1], (name, len(out), len(ref2[1]))               # everything before the bad record, nothing after it
        assert all(os.path.getsize(pat.replace("%r", str(r))) == 0 for r in (1, 2, 3))
    # a clipper run whose reads stop being of one length inside part 2: the attempt is abandoned, the one-stream run goes serial where it must
    head = text[:60_000_000]
    lines = head[:head.rindex(b"\n@")].split(b"\n")                      # whole records (the generator's quality lines never start with '@')
    assert len(lines) % 4 == 0
    cut = len(lines) // 8 * 4
    for i in range(cut, len(lines), 8):
        lines[i + 1] = lines[i + 1][:61]; lines[i + 3] = lines[i + 3][:61]
    data = b"\n".join(lines) + b"\n"
    inp.write_bytes(data)
    ad = "AGATCGGAAGAGC"
    ref = _run([REF, "fastx_clipper", "-a", ad, "-l", "15", "-v"], data)
    pat = str(tmp_path / "clip.%r.fq")
    p = subprocess.run([os.path.join(tools, "fastx_clipper"), "-a", ad, "-l", "15", "-v", "-i", str(inp), "-o", pat], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env=dict(os.environ, FXH_PARTS="4", FXH_TIMING="1"), timeout=120)
    assert p.returncode == 0 and b"fxh timing parts: abandoned" in p.stderr and b"one aligner with history from there on" in p.stderr, p.stderr[-600:]
    assert p.stdout == ref[2]                                             # the -v report (on stdout when -o names a file) is the reference's
    assert b"".join(open(pat.replace("%r", str(r)), "rb").read() for r in range(4)) == ref[1]



def test_one_output_file_by_many_strands_on_the_real_engine(tools, tmp_path):
    """`tool -i in.fq -o out.fq`, no environment that a user would set, on a tmpfs (the gated mapping: fallocate and parallel copies taking turns) and on
    the test directory's file system (positional writes): the many-strand run (host/fxh_strands.c) writes the bytes of the one-stream loop and prints its
    report, for forward slices, packed (reverse-complemented) output and the clipper; damaged input abandons the attempt on the real runtime -- contexts
    destroyed, file emptied, one stream from the top -- with the reference's exit code, message and partial output."""
    if PARSE_ENV:
        pytest.skip("the many-strand run is the device text path")
    shm = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    text = fo.synth_fastq(2, 0, 400000, 150, False)                       # 128 MB
    clip_text = fo.synth_fastq(3, 0, 300000, 100, True)
    import tempfile
    for where in ([shm] if shm else []) + [str(tmp_path)]:
        with tempfile.TemporaryDirectory(dir=where) as td:
            inp = os.path.join(td, "in.fq")
            for i, (argv, data) in enumerate([(["fastq_quality_trim_filter", "-t", "20", "-l", "30", "-q", "20", "-p", "80", "-v"], text), (["fastx_reverse_complement", "-v"], text),
                                              (["fastx_clipper", "-a", "AGATCGGAAGAGC", "-l", "15", "-v"], clip_text)]):
                open(inp, "wb").write(data)
                single, multi = os.path.join(td, "single%d" % i), os.path.join(td, "multi%d" % i)
                w = _run([os.path.join(tools, argv[0])] + argv[1:] + ["-i", inp, "-o", single], b"", env=dict(os.environ, FXH_ONE_FILE="0"))
                g = _run([os.path.join(tools, argv[0])] + argv[1:] + ["-i", inp, "-o", multi], b"", env=dict(os.environ, FXH_ONE_FILE_MIN_MB="0", FXH_STRAND_MB="4", FXH_TIMING="1"))
                assert w[0] == 0 and g[0] == 0, g[2][-500:]
                assert (b"sink gated mapping" if where == shm else b"sink pwrite") in g[2], g[2][-500:]
                assert g[1] == w[1] and open(multi, "rb").read() == open(single, "rb").read(), (where, argv)
                # the rank path with a world of one: arena on the device, the REAL ncclAllGather, drain at offset 0
                r = _run([os.path.join(tools, argv[0])] + argv[1:] + ["-i", inp, "-o", multi + ".r"], b"", env=dict(os.environ, FXH_RANK_MODE="1", FXH_STRAND_MB="4", FXH_TIMING="1"))
                assert r[0] == 0 and b"fxh timing rank 0 of 1" in r[2], r[2][-500:]
                assert r[1] == w[1] and open(multi + ".r", "rb").read() == open(single, "rb").read(), (where, argv)
            k0 = text.index(b"\n@", int(len(text) * 0.6)) + 1
            open(inp, "wb").write(text[:k0] + b"#" + text[k0 + 1:])
            argv = ["fastq_quality_trim_filter", "-t", "20", "-l", "30", "-q", "20", "-p", "80", "-v"]
            w = _run([os.path.join(tools, argv[0])] + argv[1:] + ["-i", inp, "-o", os.path.join(td, "bs")], b"", env=dict(os.environ, FXH_ONE_FILE="0"))
            g = _run([os.path.join(tools, argv[0])] + argv[1:] + ["-i", inp, "-o", os.path.join(td, "bm")], b"", env=dict(os.environ, FXH_ONE_FILE_MIN_MB="0", FXH_STRAND_MB="4", FXH_TIMING="1"))
            assert w[0] == 1 and g[0] == 1 and b"fxh timing one file: abandoned, contexts destroyed, output emptied" in g[2]
            assert _msg(g[2]) .splitlines()[-1] == _msg(w[2]).splitlines()[-1] and open(os.path.join(td, "bm"), "rb").read() == open(os.path.join(td, "bs"), "rb").read()


def test_tool_survives_a_scan_timeout(tools, tmp_path):
    """The same at the level of a command line: fastx_reverse_complement and fastq_masker (the tools whose output comes from the engine's compaction) with every
    compacting launch timing out: exit code 0, the bytes of the undisturbed run, one line on stderr that says what happened."""
    text = fo.synth_fastq(2, 0, 120000, 150, False)
    inp = tmp_path / "in.fq"
    inp.write_bytes(text)
    for argv in (["fastx_reverse_complement", "-v"], ["fastq_masker", "-q", "20", "-v"]):
        w = _run([os.path.join(tools, argv[0])] + argv[1:] + ["-i", str(inp), "-o", str(tmp_path / "calm.fq")], b"")
        g = _run([os.path.join(tools, argv[0])] + argv[1:] + ["-i", str(inp), "-o", str(tmp_path / "shaken.fq")], b"", env=dict(os.environ, FXG_TEST_SCAN_TIMEOUT="1", FXH_READ_BUFFER_MB="8"))
        assert w[0] == 0 and g[0] == 0, g[2][-400:]
        assert g[1] == w[1] and (tmp_path / "shaken.fq").read_bytes() == (tmp_path / "calm.fq").read_bytes()
        assert g[2].count(b"the GPU stopped running this process's work for a while") == 1 and b"stopped running" not in w[2]
