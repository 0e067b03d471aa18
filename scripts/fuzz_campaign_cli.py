#!/usr/bin/env python3
"""CPU: a timed random campaign of the command-line tools (host layer over the emulation stub, tests/emu/stub) against the real libfastx driver
(oracle/_ref/fxref): random tool and flags, FASTQ text with ragged stretches at random places, N, CRLF, small read buffers, lanes, sharded runs.
`python scripts/fuzz_campaign_cli.py <seed> <seconds>`; not a test, a tool for hunting what the fixed-seed tiers miss."""
import os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import emu_py
from oracle import fxoracle_py as fo
REF = fo.ref_binary()
assert REF, "oracle/_ref/fxref not built"
REAL = os.environ.get("FXG_CAMPAIGN_REAL") == "1"      # on a GPU box: the tools over the real engine (their rpath finds ../../libfxg.so)
STUB = None if REAL else emu_py.build_stub()
BIN = os.path.join(ROOT, "fastx_toolkit_amd", "host", "bin")
FAKE_RCCL = emu_py.build_fake_rccl()                     # rank-mode jobs of the campaign: the product's transport over the test-only library
if STUB: os.environ["LD_LIBRARY_PATH"] = STUB
else: os.environ.setdefault("LD_LIBRARY_PATH", "")
rng = np.random.default_rng(int(sys.argv[1]))
AD = ["AGATCGGAAGAGC", "CCTTAAGG", "TGGAATTCTCGGGTGCCAAGGAACTCCAGTCAC", "ANNTCGNA"]

def run(cmd, data, env=None):
    e = dict(os.environ, FXH_THREADS="3")
    if STUB and not (env and "LD_LIBRARY_PATH" in env): e["LD_LIBRARY_PATH"] = STUB
    e.update(env or {})
    p = subprocess.run(cmd, input=data, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e, timeout=300)
    return p.returncode, p.stdout, p.stderr

def text(n, L, ad):
    lines = fo.synth_fastq(int(rng.integers(1, 1 << 30)), 0, n, L, True).split(b"\n")
    recs = [lines[4 * i:4 * i + 4] for i in range(n)]
    a, b = sorted(int(x) for x in rng.integers(0, n + 1, size=2))
    mode = int(rng.integers(0, 4))                       # 0 fixed, 1 ragged in [a, b), 2 ragged everywhere, 3 ragged from a on
    out = []
    for i, l in enumerate(recs):
        rag = (mode == 1 and a <= i < b) or mode == 2 or (mode == 3 and i >= a)
        k = int(rng.integers(1, L + 1)) if rag and rng.random() < 0.6 else L
        out.append(b"\n".join([l[0], l[1][:k], l[2], l[3][:k]]))
    form = rng.random()
    collapsed = form < 0.12 and rng.random() < 0.4
    if form < 0.12:                                      # FASTA (collapsed-read identifiers in some files): the tools that take it
        out = [b">" + (b"%d-%d" % (i, int(rng.integers(1, 50))) if collapsed and rng.random() < 0.5 else l.split(b"\n")[0][1:]) + b"\n" + l.split(b"\n")[1] for i, l in enumerate(out)]
    elif form < 0.2:                                     # numeric quality lines
        out = [b"\n".join([l.split(b"\n")[0], l.split(b"\n")[1], l.split(b"\n")[2], b" ".join(b"%d" % (c - 33) for c in l.split(b"\n")[3])]) for l in out]
    t = b"\n".join(out) + b"\n"
    return (t.replace(b"\n", b"\r\n") if rng.random() < 0.1 else t), form < 0.12, collapsed

t0 = time.time(); n = 0
with tempfile.TemporaryDirectory() as tmp:
    while time.time() - t0 < float(sys.argv[2]):
        ad = AD[int(rng.integers(0, len(AD)))]
        L = int(rng.choice([36, 50, 100, 150, 251, 300]))
        data, fasta, collapsed = text(int(rng.integers(1, 30000)), L, ad)
        # FASTA: clipper, reverse-complement, fixed trimmer, artifacts filter, statistics -- the statistics tool not on collapsed reads: its percentile walk then leaves the
        # reference's static array and prints whatever the linker put behind it
        tool = int(rng.choice([0, 2, 5, 7] if collapsed else [0, 2, 5, 7, 9])) if fasta else int(rng.integers(0, 10))
        clipf = ["-a", ad, "-l", str(int(rng.integers(0, 30)))] + [f for f in ("-n", "-c", "-C", "-k") if rng.random() < 0.25] + (["-M", str(int(rng.integers(1, 12)))] if rng.random() < 0.3 else [])
        if "-c" in clipf and "-C" in clipf: clipf.remove("-C")
        q = ["-t", str(int(rng.integers(1, 40))), "-l", str(int(rng.integers(0, 60)))]
        if tool == 0: chain = [["fastx_clipper"] + clipf + ["-v"]]; fused = chain[0]
        elif tool == 1: chain = [["fastq_quality_trimmer"] + q + ["-v"]]; fused = chain[0]
        elif tool == 2: chain = [["fastx_reverse_complement"]]; fused = chain[0]
        elif tool == 4: chain = [["fastq_quality_filter", "-q", str(int(rng.integers(0, 42))), "-p", str(int(rng.integers(1, 101))), "-v"]]; fused = chain[0]
        elif tool == 5: chain = [["fastx_trimmer", "-f", str(int(rng.integers(1, 30))), "-l", str(int(rng.integers(30, 300))), "-v"] if rng.random() < 0.5 else ["fastx_trimmer", "-t", str(int(rng.integers(1, 30))), "-m", str(int(rng.integers(1, 60))), "-v"]]; fused = chain[0]
        elif tool == 6: chain = [["fastq_masker", "-q", str(int(rng.integers(0, 45))), "-r", str(rng.choice(list("N.x"))), "-v"]]; fused = chain[0]
        elif tool == 7: chain = [["fastx_artifacts_filter", "-v"]]; fused = chain[0]
        elif tool == 8: chain = [["fastq_to_fasta", "-v"] + (["-r"] if rng.random() < 0.5 else []) + (["-n"] if rng.random() < 0.5 else [])]; fused = chain[0]
        elif tool == 9: chain = [["fastx_quality_stats"] + (["-N"] if rng.random() < 0.5 and not fasta else [])]; fused = chain[0]      # (-N on FASTA: the last column's per-class quartiles come from behind the reference's static array)
        else:
            qq, pp = str(int(rng.integers(0, 40))), str(int(rng.integers(1, 101)))
            cf = [f for f in clipf if f not in ("-c", "-C", "-k")]
            chain = [["fastx_clipper"] + cf, ["fastq_quality_trimmer", "-t", q[1], "-l", q[3]], ["fastq_quality_filter", "-q", qq, "-p", pp]]
            fused = ["fastx_clip_trim_filter"] + cf + ["-t", q[1], "-m", q[3], "-q", qq, "-p", pp]
        want = data; rc = 0; err = b""; skip = False
        for i, c in enumerate(chain):
            rc, want, err = run([REF] + c, want)
            if rc: break
            if not want and i + 1 < len(chain): skip = True; break      # an empty intermediate file is an error to the reference's next tool, not to the one-pass tool
        if skip: continue
        env = {"FXH_READ_BUFFER_MB": str(int(rng.choice([1, 2, 8])))}
        if rng.random() < 0.4: env["FXH_LANES"] = str(int(rng.integers(1, 5)))
        if rng.random() < 0.3 and tool not in (9,):      # file to file, sharded
            inp, pat = os.path.join(tmp, "in.fq"), os.path.join(tmp, "o.%r.fq")
            open(inp, "wb").write(data)
            k = int(rng.integers(2, 5))
            got = run([os.path.join(BIN, fused[0])] + [f for f in fused[1:] if f != "-v"] + ["-i", inp, "-o", pat], b"", dict(env, FXH_PARTS=str(k)))
            files = [pat.replace("%r", str(r)) for r in range(k)]
            out = b"".join(open(f, "rb").read() for f in files if os.path.exists(f))
            for f in files:
                if os.path.exists(f): os.unlink(f)
            ok = (got[0], out) == (rc, want) if rc == 0 else got[0] == rc
        elif rng.random() < 0.45 and tool not in (9,):    # file to ONE file by many strands (round 5: tickets, published sizes, both sinks), or as a job of 2-4 rank processes
            inp, outp = os.path.join(tmp, "in.fq"), os.path.join(tmp, "one.fq")
            open(inp, "wb").write(data)
            e1 = dict(env, FXH_ONE_FILE_MIN_MB="0", FXH_STRAND_KB=str(int(rng.choice([4, 16, 64, 256, 1024]))), FXH_STRANDS=str(int(rng.integers(1, 6))),
                      FXH_ONE_FILE_SINK=str(rng.choice(["map", "pwrite"])), FXH_ONE_FILE_WINDOW_MB="1", FXH_STRAND_OUT_SLOTS=str(int(rng.integers(2, 5))))
            argv = [os.path.join(BIN, fused[0])] + fused[1:] + ["-i", inp, "-o", outp]
            if rng.random() < 0.35:
                world = int(rng.integers(2, 5))
                e1.update(FXH_WORLD=str(world), FXH_DRAIN_MB="1", LD_LIBRARY_PATH=FAKE_RCCL + os.pathsep + os.environ["LD_LIBRARY_PATH"], FXH_THREADS="3", **({"FXG_FAKE_RCCL_HIP": "1"} if REAL else {}))
                ps = [subprocess.Popen(argv, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **dict(e1, FXH_RANK=str(r)))) for r in range(world)]
                res = [(p.communicate(timeout=300), p.returncode) for p in ps]
                got = (res[0][1], res[0][0][0], res[0][0][1])
                ok_others = all(c == 0 and o == b"" for (o, _), c in res[1:])
            else:
                got = run(argv, b"", e1); ok_others = True
            out = open(outp, "rb").read() if os.path.exists(outp) else b""
            if os.path.exists(outp): os.unlink(outp)
            env = e1
            # (-v reports go to stdout when -o names a file: compare them with the reference's stderr report when there is one tool)
            ok = ok_others and ((got[0], out) == (rc, want) if rc == 0 else (got[0] == rc and got[2].split(b": ", 1)[-1].splitlines()[-1:] == err.split(b": ", 1)[-1].splitlines()[-1:]))
            got = (got[0], out, got[2])
        else:
            got = run([os.path.join(BIN, fused[0])] + fused[1:], data, env)
            ok = (got[0], got[1]) == (rc, want) and (len(chain) > 1 or got[2].split(b": ", 1)[-1] == err.split(b": ", 1)[-1] or rc == 0)
        if not ok:
            open("/tmp/fuzz_cli_fail.fq", "wb").write(data)
            raise SystemExit("MISMATCH seed %s case %d: %r env %r rc %r/%r out %d/%d bytes; input kept in /tmp/fuzz_cli_fail.fq\n%s" % (sys.argv[1], n, fused, env, got[0], rc, len(got[1]), len(want), got[2][-400:].decode(errors="replace")))
        n += 1
print("seed", sys.argv[1], "cases", n, "ok")
