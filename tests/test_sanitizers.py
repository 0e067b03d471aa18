"""Sanitizer tier for the threaded host C layer (the reference builds -Wall -Wextra -Werror, configure.ac:34).

`make -C fastx_toolkit_amd/host SAN=address,undefined` and `SAN=thread` build instrumented copies of the tools (bin_san_*/); they run
against the emulation stub (no GPU needed) over the paths where a latent race or out-of-bounds access would live: reader threads,
lanes, the sharded run and its restart, the asynchronous writer, parallel deflate, the host parser's worker threads -- on well-formed
input and on hand-written malformed / corner-case input.  Every run must (1) print no sanitizer report and (2) behave exactly like
the plain build (exit code, stdout, message).
"""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from oracle import fxoracle_py as fo

pytestmark = pytest.mark.sanitizers
HOST = os.path.join(ROOT, "fastx_toolkit_amd", "host")
STUB_DIR = os.path.join(ROOT, "tests", "emu", "stub")
REPORT = (b"ERROR: AddressSanitizer", b"runtime error:", b"WARNING: ThreadSanitizer", b"ERROR: LeakSanitizer", b"SUMMARY: UndefinedBehaviorSanitizer")


def _corner_inputs():
    rec = b"@r1\nACGTACGTAC\n+\nIIIIIIIIII\n"
    long_id = b"@" + b"x" * 30000 + b"\nACGT\n+\nIIII\n"
    return {
        "empty": b"", "only_newline": b"\n", "just_at": b"@", "header_only": b"@r1\n", "two_lines": b"@r1\nACGT\n", "three_lines": b"@r1\nACGT\n+\n",
        "no_final_newline": rec[:-1], "truncated_quality": b"@r1\nACGTACGT\n+\nIII\n", "long_quality": b"@r1\nACGT\n+\nIIIIIIII\n",
        "bad_base": b"@r1\nACGXT\n+\nIIIII\n", "lowercase": b"@r1\nacgt\n+\nIIII\n", "empty_sequence": b"@r1\n\n+\n\n", "no_plus": b"@r1\nACGT\n-\nIIII\n",
        "crlf": rec.replace(b"\n", b"\r\n") * 3, "cr_only": rec.replace(b"\n", b"\r"), "mixed_numeric": rec + b"@r2\nACGT\n+\n40 40 30 2\n" + rec,
        "numeric_overflow": b"@r1\nAC\n+\n99999999999999999999 3\n", "numeric_negative": b"@r1\nACG\n+\n-5 -15 -16\n", "numeric_junk": b"@r1\nACG\n+\n3 x 4\n",
        "quality_too_high": b"@r1\nACGT\n+\nII\x7fI\n", "quality_too_low": b"@r1\nACGT\n+\nII\x10I\n", "high_bit": b"@r1\nACGT\n+\nII\xffI\n",
        "fasta": b">1-5\nACGTN\n>2\nTTGCA\n", "fasta_in_fastq": rec + b">x\nACGT\n", "long_id": long_id, "id_24999": b"@" + b"i" * 24998 + b"\nACGT\n+\nIIII\n",
        "blank_between": rec + b"\n" + rec, "garbage": bytes(range(256)) * 4, "all_tilde": b"@r\nACGT\n+\n~~~~\n", "many_small": b"@\nA\n\nI\n" * 5000,
        "second_record_bad": rec * 50 + b"@bad\nACGT\n+\nII\n" + rec * 50,
    }


def _build(san):
    subprocess.check_call(["make", "-s", "-C", HOST, "SAN=" + san])
    return os.path.join(HOST, "bin_san_" + san.replace(",", "_"))


def _run(cmd, data, env_extra, san_env):
    env = dict(os.environ, LD_LIBRARY_PATH=STUB_DIR, FXH_THREADS="4", **san_env)
    env.update(env_extra)
    p = subprocess.run(cmd, input=data, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=300)
    return p.returncode, p.stdout, p.stderr


def _msg(err):
    lines = [l for l in err.splitlines() if not l.startswith(b"fxh timing")]
    return b"\n".join(l.split(b": ", 1)[-1] for l in lines)


def _plain_build():
    import emu_py
    from fastx_toolkit_amd import build as b
    b.build_engine()
    emu_py.build_stub()
    subprocess.check_call(["make", "-s", "-C", HOST])
    return os.path.join(HOST, "bin")


@pytest.mark.parametrize("san", ["address,undefined", "thread"])
def test_tools_under_sanitizers(san, tmp_path):
    plain_bin = _plain_build()
    san_bin = _build(san)
    san_env = ({"ASAN_OPTIONS": "detect_leaks=0:abort_on_error=0:exitcode=99", "UBSAN_OPTIONS": "print_stacktrace=1"} if "address" in san
               else {"TSAN_OPTIONS": "exitcode=99:report_signal_unsafe=0"})
    rng = np.random.default_rng(91)
    text = fo.synth_fastq(61, 0, 24000, 100, True)                     # ~5.6 MB: several 1 MB blocks
    big = tmp_path / "in.fq"
    big.write_bytes(text)
    ad = "AGATCGGAAGAGC"
    runs = []                                                            # (argv, stdin, env)
    for env in ({"FXH_READ_BUFFER_MB": "1"}, {"FXH_READ_BUFFER_MB": "1", "FXH_LANES": "3"}, {"FXH_READ_BUFFER_MB": "1", "FXH_HOST_PARSE": "1"},
                {"FXH_READ_BUFFER_MB": "1", "FXG_EMU_NO_TEXT": "1"}, {"FXH_READ_BUFFER_MB": "2", "FXH_NO_OVERLAP": "1"},
                {"FXH_READ_BUFFER_MB": "1", "FXG_EMU_DEVICES": "2", "FXG_DEVICES": "0,1"}):
        runs.append((["fastq_quality_trim_filter", "-t", "20", "-l", "30", "-q", "20", "-p", "80", "-v"], text, env))
    runs += [(["fastx_clipper", "-a", ad, "-l", "15", "-n", "-v"], text[:1500000], {"FXH_READ_BUFFER_MB": "1"}),
             (["fastx_reverse_complement"], text[:2000000], {"FXH_READ_BUFFER_MB": "1"}),
             (["fastq_masker", "-q", "20", "-z"], text[:2000000], {"FXH_READ_BUFFER_MB": "1"}),
             (["fastx_quality_stats"], text[:1000000], {"FXH_READ_BUFFER_MB": "1"}),
             (["fastq_to_fasta", "-r"], text[:1000000], {}),
             # files: parallel pread, positional writes, the sharded run, and its restart on a damaged record
             (["fastq_quality_trimmer", "-t", "20", "-l", "30", "-i", str(big), "-o", str(tmp_path / "o1.fq")], b"", {"FXH_READ_BUFFER_MB": "1", "FXH_IO_THREADS": "3"}),
             (["fastq_quality_trimmer", "-t", "20", "-l", "30", "-v", "-i", str(big), "-o", str(tmp_path / "p.%r.fq")], b"", {"FXH_READ_BUFFER_MB": "1", "FXH_PARTS": "3"})]
    k0 = text.index(b"\n@", int(len(text) * 0.7)) + 1
    bad = tmp_path / "bad.fq"
    bad.write_bytes(text[:k0] + b"#" + text[k0 + 1:])
    runs.append((["fastq_quality_trimmer", "-t", "20", "-l", "30", "-i", str(bad), "-o", str(tmp_path / "b.%r.fq")], b"", {"FXH_READ_BUFFER_MB": "1", "FXH_PARTS": "3"}))
    # the clipper's automatic mode: lanes in parallel while the reads have one length, the hand-over to one aligner (fxh_clip_go_serial) at the first
    # ragged block -- here half way through -- and a sharded clipper run that is kept (fixed-length input) / started over as one stream (ragged part);
    # `-o NAME.%r.fq` without FXH_PARTS (the tool picks the number of parts itself: one, for an input of this size)
    lines = text.split(b"\n")
    recs = [lines[4 * i:4 * i + 4] for i in range(len(lines) // 4)]
    half = len(recs) // 2
    rag_text = b"".join(b"\n".join(l) + b"\n" for l in recs[:half])
    for l in recs[half:]:
        L = int(rng.integers(20, 100))
        rag_text += b"\n".join([l[0], l[1][:L], l[2], l[3][:L]]) + b"\n"
    rag = tmp_path / "ragged.fq"
    rag.write_bytes(rag_text)
    runs += [(["fastx_clipper", "-a", ad, "-l", "15", "-v"], rag_text, {"FXH_READ_BUFFER_MB": "1", "FXH_LANES": "3"}),
             (["fastx_clip_trim_filter", "-a", ad, "-l", "15", "-n", "-t", "20", "-m", "30", "-q", "20", "-p", "80", "-v"], rag_text, {"FXH_READ_BUFFER_MB": "1"}),
             (["fastx_clipper", "-a", ad, "-l", "15", "-v", "-i", str(big), "-o", str(tmp_path / "c.%r.fq")], b"", {"FXH_READ_BUFFER_MB": "1", "FXH_PARTS": "3"}),
             (["fastx_clipper", "-a", ad, "-l", "15", "-v", "-i", str(rag), "-o", str(tmp_path / "cr.%r.fq")], b"", {"FXH_READ_BUFFER_MB": "1", "FXH_PARTS": "3"}),
             (["fastq_quality_trimmer", "-t", "20", "-l", "30", "-v", "-i", str(big), "-o", str(tmp_path / "auto.%r.fq")], b"", {"FXH_READ_BUFFER_MB": "1"})]
    # round 5: ONE output file written by many strands (tickets, published sizes, the gated allocator and the copy pool; positional writes), its
    # abandon path on a damaged record and on ragged clipper input, the rank path with a world of one (arena, all-gather over the test transport, drain)
    import emu_py
    one = {"FXH_ONE_FILE_MIN_MB": "0", "FXH_STRAND_KB": "256", "FXH_STRANDS": "3", "FXH_ONE_FILE_WINDOW_MB": "1", "FXH_READ_BUFFER_MB": "1"}
    for sink in ("map", "pwrite"):
        runs += [(["fastq_quality_trim_filter", "-t", "20", "-l", "30", "-q", "20", "-p", "80", "-v", "-i", str(big), "-o", str(tmp_path / ("one_%s.fq" % sink))], b"", dict(one, FXH_ONE_FILE_SINK=sink)),
                 (["fastx_reverse_complement", "-v", "-i", str(big), "-o", str(tmp_path / ("one_rc_%s.fq" % sink))], b"", dict(one, FXH_ONE_FILE_SINK=sink, FXG_EMU_DEVICES="2", FXG_DEVICES="0,1")),
                 (["fastq_quality_trimmer", "-t", "20", "-l", "30", "-i", str(bad), "-o", str(tmp_path / ("one_bad_%s.fq" % sink))], b"", dict(one, FXH_ONE_FILE_SINK=sink)),
                 (["fastx_clipper", "-a", ad, "-l", "15", "-v", "-i", str(rag), "-o", str(tmp_path / ("one_rag_%s.fq" % sink))], b"", dict(one, FXH_ONE_FILE_SINK=sink))]
    runs.append((["fastq_quality_trim_filter", "-t", "20", "-l", "30", "-q", "20", "-p", "80", "-v", "-i", str(big), "-o", str(tmp_path / "rankmode.fq")], b"",
                 dict(one, FXH_RANK_MODE="1", FXH_DRAIN_MB="1", LD_LIBRARY_PATH=STUB_DIR + os.pathsep + emu_py.build_fake_rccl())))
    # ... and with the sink a tmpfs gets: rank 0's allocator for the job's pages, the barrier exchange, copies into a mapping of the slice, the watch threads
    runs.append((["fastx_reverse_complement", "-v", "-i", str(big), "-o", str(tmp_path / "rankmode_paged.fq")], b"",
                 dict(one, FXH_RANK_MODE="1", FXH_DRAIN_MB="1", FXH_ONE_FILE_SINK="map", LD_LIBRARY_PATH=STUB_DIR + os.pathsep + emu_py.build_fake_rccl())))
    for name, data in _corner_inputs().items():
        for argv in (["fastq_quality_trimmer", "-t", "20", "-l", "2"], ["fastx_trimmer", "-f", "2", "-l", "9"], ["fastx_reverse_complement"]):
            runs.append((argv, data, {"FXH_READ_BUFFER_MB": "1"} if len(data) % 2 else {}))
        runs.append((["fastq_quality_filter", "-q", "20", "-p", "50"], data, {"FXH_HOST_PARSE": "1"}))
    checked = 0
    for argv, data, env in runs:
        want = _run([os.path.join(plain_bin, argv[0])] + argv[1:], data, env, {})
        got = _run([os.path.join(san_bin, argv[0])] + argv[1:], data, env, san_env)
        for marker in REPORT:
            assert marker not in got[2], (san, argv, env, got[2][-1500:].decode(errors="replace"))
        assert got[0] != 99, (san, argv, env, got[2][-800:])
        assert (got[0], got[1]) == (want[0], want[1]) and _msg(got[2]) == _msg(want[2]), (san, argv, env, got[2][-300:], want[2][-300:])
        checked += 1
    assert checked > 100
