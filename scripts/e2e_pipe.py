#!/usr/bin/env python3
"""`fastq_quality_trimmer -t 20 -l 30 -i in.fq | fastq_quality_filter -q 20 -p 80 -o out.fq` on the GPU box, FASTQ on tmpfs (not a test): the pipe as the
reference's users type it, two GPU processes.  READS (default 16 M); MATRIX = comma-separated environments (':'-separated ENV=VAL, empty = none)."""
import hashlib
import os
import subprocess
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import fxoracle_py as fo  # noqa: E402

reads = int(os.environ.get("READS", "16000000"))
matrix = os.environ.get("MATRIX", ",FXH_NO_PIPE_FANOUT=1,FXH_NO_PIPE_TUNING=1:FXH_NO_PIPE_FANOUT=1").split(",")
bindir = os.path.join(ROOT, "fastx_toolkit_amd", "host", "bin")
chunk = 250_000
with tempfile.TemporaryDirectory(dir="/dev/shm") as td:
    inp, out = os.path.join(td, "in.fq"), os.path.join(td, "out.fq")
    with open(inp, "wb") as f:
        with ThreadPoolExecutor(max_workers=min(os.cpu_count() or 2, 48)) as ex:
            for part in ex.map(lambda k: fo.synth_fastq(2, k * chunk, chunk, 150, False), range(reads // chunk)):
                f.write(part)
    ref = None
    for item in matrix:
        env = dict(os.environ, FXH_TIMING="1")
        for kv in [x for x in item.split(":") if x]:
            env[kv.split("=")[0]] = kv.split("=", 1)[1]
        walls, err = [], b""
        for rep in range(3):
            if os.path.exists(out):
                os.unlink(out)
            t0 = time.perf_counter()
            p1 = subprocess.Popen([os.path.join(bindir, "fastq_quality_trimmer"), "-t", "20", "-l", "30", "-i", inp], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
            p2 = subprocess.Popen([os.path.join(bindir, "fastq_quality_filter"), "-q", "20", "-p", "80", "-o", out], stdin=p1.stdout, stderr=subprocess.PIPE, env=env)
            p1.stdout.close()
            e2 = p2.communicate()[1]
            e1 = p1.stderr.read()
            assert p1.wait() == 0 and p2.returncode == 0, (e1[-300:], e2[-300:])
            walls.append(time.perf_counter() - t0)
            err = e1 + e2
        h = hashlib.md5()
        with open(out, "rb") as f:
            for blk in iter(lambda: f.read(1 << 24), b""):
                h.update(blk)
        ref = ref or h.hexdigest()
        best = min(walls)
        print("[%s] wall %s s  %.1f Mreads/s  %.2f Gbases/s  md5 %s" % (item or "no environment", " ".join("%.3f" % w for w in walls), reads / best / 1e6, reads * 150 / best / 1e9,
                                                                     "same" if h.hexdigest() == ref else "DIFFERS"), flush=True)
        for l in err.decode(errors="replace").splitlines():
            if l.startswith("fxh timing part"):
                print("    " + l[:400], flush=True)
