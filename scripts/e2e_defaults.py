#!/usr/bin/env python3
"""GPU box: the UN-TUNED command lines end to end (FASTQ text on tmpfs in and out, no environment variable): what a user who knows no
knob gets (round-3 verdict item 4c).  fastx_clipper and fastx_clip_trim_filter on fixed-length and on ragged input (ragged from 50 % on),
fastq_quality_trim_filter with `-o out.fq` and with `-o out.%r.fq` (the tool picks the parts).  One JSON line per command."""
import json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from concurrent.futures import ThreadPoolExecutor
from oracle import fxoracle_py as fo
BIN = os.path.join(ROOT, "fastx_toolkit_amd", "host", "bin")
READS = int(os.environ.get("READS", "16000000"))
AD = "AGATCGGAAGAGC"
chunk = 250_000


def gen(path, seed, ragged_from=None):
    rng = np.random.default_rng(3)
    with open(path, "wb") as f, ThreadPoolExecutor(max_workers=32) as ex:
        for k, part in enumerate(ex.map(lambda k: fo.synth_fastq(seed, k * chunk, chunk, 150, True), range(READS // chunk))):
            if ragged_from is not None and k * chunk >= ragged_from * READS:
                lines = part.split(b"\n")
                for i in range(0, len(lines) - 1, 8):                     # every other record shortened
                    L = int(rng.integers(40, 150))
                    lines[i + 1] = lines[i + 1][:L]; lines[i + 3] = lines[i + 3][:L]
                part = b"\n".join(lines)
            f.write(part)


def timed(name, cmd, outs, env=None):
    best = None
    for _ in range(2):
        for o in outs:
            if os.path.exists(o):
                os.unlink(o)
        t0 = time.perf_counter()
        p = subprocess.run(cmd, env=env or os.environ, stderr=subprocess.PIPE)
        dt = time.perf_counter() - t0
        if p.returncode != 0:
            print(json.dumps(dict(name=name, error=p.stderr.decode()[-300:])))
            return
        best = dt if best is None else min(best, dt)
    serial = b"one aligner" in p.stderr
    print(json.dumps(dict(name=name, command=" ".join(os.path.basename(c) if i == 0 else (c if not c.startswith("/") else os.path.basename(c)) for i, c in enumerate(cmd)),
                          reads=READS, wall_s=round(best, 3), mreads_s=round(READS / best / 1e6, 2), gbases_in_s=round(os.path.getsize(cmd[cmd.index("-i") + 1]) * 150 / 320 / best / 1e9, 2),
                          went_serial=serial, output_bytes=sum(os.path.getsize(o) for o in outs if os.path.exists(o)))), flush=True)


with tempfile.TemporaryDirectory(dir="/dev/shm") as td:
    fixed, rag = os.path.join(td, "fixed.fq"), os.path.join(td, "ragged.fq")
    gen(fixed, 5)
    gen(rag, 5, 0.5)
    o1 = os.path.join(td, "out.fq")
    parts = [os.path.join(td, "out.%d.fq" % r) for r in range(4)] + [os.path.join(td, "out.parts.fq")]
    tenv = dict(os.environ, FXH_TIMING="1")
    timed("fastx_clipper, fixed-length input", [os.path.join(BIN, "fastx_clipper"), "-a", AD, "-l", "15", "-n", "-i", fixed, "-o", o1], [o1], tenv)
    timed("fastx_clipper, fixed-length input, -o out.%r.fq", [os.path.join(BIN, "fastx_clipper"), "-a", AD, "-l", "15", "-n", "-i", fixed, "-o", os.path.join(td, "out.%r.fq")], parts, tenv)
    timed("fastx_clipper, ragged from 50 %", [os.path.join(BIN, "fastx_clipper"), "-a", AD, "-l", "15", "-n", "-i", rag, "-o", o1], [o1], tenv)
    timed("fastx_clipper, ragged from 50 %, FXH_CLIP_SERIAL=1 (round 3's default)", [os.path.join(BIN, "fastx_clipper"), "-a", AD, "-l", "15", "-n", "-i", rag, "-o", o1], [o1], dict(tenv, FXH_CLIP_SERIAL="1"))
    timed("fastx_clipper, fixed-length input, FXH_CLIP_SERIAL=1 (round 3's default)", [os.path.join(BIN, "fastx_clipper"), "-a", AD, "-l", "15", "-n", "-i", fixed, "-o", o1], [o1], dict(tenv, FXH_CLIP_SERIAL="1"))
    c5 = [os.path.join(BIN, "fastx_clip_trim_filter"), "-a", AD, "-l", "15", "-n", "-t", "20", "-m", "30", "-q", "20", "-p", "80"]
    timed("fastx_clip_trim_filter (config 5 in one pass), fixed-length input", c5 + ["-i", fixed, "-o", o1], [o1], tenv)
    timed("fastx_clip_trim_filter, fixed-length input, -o out.%r.fq", c5 + ["-i", fixed, "-o", os.path.join(td, "out.%r.fq")], parts, tenv)
    timed("fastx_clip_trim_filter, ragged from 50 %", c5 + ["-i", rag, "-o", o1], [o1], tenv)
    tf = [os.path.join(BIN, "fastq_quality_trim_filter"), "-t", "20", "-l", "30", "-q", "20", "-p", "80"]
    timed("fastq_quality_trim_filter", tf + ["-i", fixed, "-o", o1], [o1], tenv)
    timed("fastq_quality_trim_filter, -o out.%r.fq", tf + ["-i", fixed, "-o", os.path.join(td, "out.%r.fq")], parts, tenv)
