#!/usr/bin/env python3
"""End-to-end matrix of the sharded tool run (FXH_PARTS) on the GPU box (not a test): FASTQ on tmpfs -> fastq_quality_trim_filter -> parts on tmpfs.
READS (default 16 M), MATRIX = "parts:lanes,..." (default 1:2,2:2,4:2,4:1,8:1)."""
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
matrix = [tuple(m.split(":")) for m in os.environ.get("MATRIX", "1:2,2:2,4:2,4:1,8:1").split(",")]     # parts:lanes[:ENV=VAL[:ENV=VAL]]
tool = os.path.join(ROOT, "fastx_toolkit_amd", "host", "bin", "fastq_quality_trim_filter")
chunk = 250_000
with tempfile.TemporaryDirectory(dir="/dev/shm") as td:
    inp = os.path.join(td, "in.fq")
    t0 = time.perf_counter()
    with open(inp, "wb") as f:
        with ThreadPoolExecutor(max_workers=min(os.cpu_count() or 2, 48)) as ex:
            for part in ex.map(lambda k: fo.synth_fastq(2, k * chunk, chunk, 150, False), range(reads // chunk)):
                f.write(part)
    print("input: %d reads, %.2f GB, generated in %.1f s" % (reads, os.path.getsize(inp) / 1e9, time.perf_counter() - t0), flush=True)
    ref_md5 = None
    for item in matrix:
        parts, lanes = int(item[0]), int(item[1])
        env = dict(os.environ, FXH_LANES=str(lanes), FXH_TIMING="1")
        for kv in item[2:]:
            env[kv.split("=")[0]] = kv.split("=")[1]
        if parts > 1:
            env["FXH_PARTS"] = str(parts)
        pat = os.path.join(td, "out.%r.fq") if parts > 1 else os.path.join(td, "out.fq")
        best, errtxt = None, b""
        for rep in range(2):
            for f in os.listdir(td):
                if f.startswith("out."):
                    os.unlink(os.path.join(td, f))
            t0 = time.perf_counter()
            p = subprocess.run([tool, "-t", "20", "-l", "30", "-q", "20", "-p", "80", "-i", inp, "-o", pat], env=env, stderr=subprocess.PIPE)
            dt = time.perf_counter() - t0
            assert p.returncode == 0, p.stderr[-500:]
            if best is None or dt < best:
                best, errtxt = dt, p.stderr
        h = hashlib.md5()
        nbytes = 0
        for r in range(parts):
            with open(pat.replace("%r", str(r)), "rb") as f:
                for blk in iter(lambda: f.read(1 << 24), b""):
                    h.update(blk); nbytes += len(blk)
        if ref_md5 is None:
            ref_md5 = h.hexdigest()
        print("parts %d lanes/part %d %s: wall %.3f s  %.1f Mreads/s  %.2f Gbases/s  out %.2f GB  md5 %s %s" % (
            parts, lanes, " ".join(item[2:]), best, reads / best / 1e6, reads * 150 / best / 1e9, nbytes / 1e9, h.hexdigest(), "== single" if h.hexdigest() == ref_md5 else "DIFFERS"), flush=True)
        for l in errtxt.decode(errors="replace").splitlines():
            if l.startswith("fxh timing"):
                print("    " + l[:330], flush=True)
