#!/usr/bin/env python3
"""End-to-end timing of the C tools on FASTQ text in tmpfs (GPU box): wall time file -> file, phase times (FXH_TIMING), md5 of the outputs.

    python scripts/e2e_cli.py [READS]          # default 16 M reads x 150 bp = 5.1 GB of text
"""
import hashlib
import json
import os
import subprocess
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import fxoracle_py as fo  # noqa: E402

READS = int(sys.argv[1]) if len(sys.argv) > 1 else 16_000_000
BIN = os.path.join(ROOT, "fastx_toolkit_amd", "host", "bin")
subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "fastx_toolkit_amd", "host")])
inp, out = "/dev/shm/e2e_in.fq", "/dev/shm/e2e_out.fq"
chunk = 250_000
t0 = time.time()
with ThreadPoolExecutor(max_workers=32) as ex:
    parts = list(ex.map(lambda k: fo.synth_fastq(2, k * chunk, chunk, 150, False), range(READS // chunk)))
with open(inp, "wb") as f:
    for p in parts:
        f.write(p)
del parts
print(json.dumps(dict(reads=READS, input_bytes=os.path.getsize(inp), gen_s=round(time.time() - t0, 1))), flush=True)


def md5(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    return h.hexdigest()


def run(label, argv, env=None, pipe_to=None):
    e = dict(os.environ, FXH_TIMING="1", **(env or {}))
    best, timing = None, ""
    for _ in range(2):
        if os.path.exists(out):
            os.unlink(out)
        t0 = time.perf_counter()
        if pipe_to:
            p1 = subprocess.Popen(argv + ["-i", inp], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e)
            p2 = subprocess.Popen(pipe_to + ["-o", out], stdin=p1.stdout, stderr=subprocess.PIPE, env=e)
            p1.stdout.close()
            e2 = p2.communicate()[1]
            e1 = p1.stderr.read()
            rc = p1.wait() or p2.returncode
            err = e1 + e2
        else:
            p = subprocess.run(argv + ["-i", inp, "-o", out], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e)
            rc, err = p.returncode, p.stderr + p.stdout
        dt = time.perf_counter() - t0
        if rc != 0:
            print(json.dumps(dict(label=label, rc=rc, err=err.decode(errors="replace")[-300:])), flush=True)
            return
        if best is None or dt < best:
            best, timing = dt, " | ".join(l for l in err.decode(errors="replace").splitlines() if l.startswith("fxh timing"))
    print(json.dumps(dict(label=label, wall_s=round(best, 3), mreads_s=round(READS / best / 1e6, 2), gbases_s=round(READS * 150 / best / 1e9, 2),
                          out_bytes=os.path.getsize(out), md5=md5(out), timing=timing)), flush=True)


T = [os.path.join(BIN, "fastq_quality_trimmer"), "-t", "20", "-l", "30"]
F = [os.path.join(BIN, "fastq_quality_filter"), "-q", "20", "-p", "80"]
TF = [os.path.join(BIN, "fastq_quality_trim_filter"), "-t", "20", "-l", "30", "-q", "20", "-p", "80"]
run("trimmer lanes=2 (default)", T)
run("trimmer lanes=1", T, {"FXH_LANES": "1"})
run("trimmer lanes=3", T, {"FXH_LANES": "3"})
run("trimmer lanes=2 io=1", T, {"FXH_IO_THREADS": "1"})
run("trimmer lanes=2 io=16 threads=32", T, {"FXH_IO_THREADS": "16", "FXH_THREADS": "32"})
run("trimmer host-parse", T, {"FXH_HOST_PARSE": "1"})
run("trimmer two contexts as two devices", T, {"FXG_DEVICES": "0,0", "FXH_LANES": "2"})
run("fused trim+filter", TF)
run("fused trim+filter lanes=3 io=16", TF, {"FXH_LANES": "3", "FXH_IO_THREADS": "16"})
run("pipe trimmer | filter", T, pipe_to=F)
run("revcomp", [os.path.join(BIN, "fastx_reverse_complement")])
run("clipper", [os.path.join(BIN, "fastx_clipper"), "-a", "AGATCGGAAGAGC", "-l", "15", "-n"])
run("clipper parallel lanes (fixed-length input)", [os.path.join(BIN, "fastx_clipper"), "-a", "AGATCGGAAGAGC", "-l", "15", "-n"], {"FXH_CLIP_PARALLEL": "1"})
os.unlink(inp)
if os.path.exists(out):
    os.unlink(out)
