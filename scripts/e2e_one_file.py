#!/usr/bin/env python3
"""End-to-end matrix of the default command line `tool -i in.fq -o out.fq` on the GPU box (not a test): FASTQ on tmpfs -> fastq_quality_trim_filter -> ONE
file on tmpfs.  READS (default 64 M); MATRIX = comma-separated runs, each a ':'-separated list of ENV=VAL (empty = the default invocation, no
environment at all); the first run is the one-stream reference (FXH_ONE_FILE=0) whose md5 the others must have.  RANKS=n in an item: a rank job of n processes sharing the GPU."""
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

reads = int(os.environ.get("READS", "64000000"))
matrix = os.environ.get("MATRIX", "FXH_ONE_FILE=0,,FXH_STRANDS=4,FXH_STRANDS=12").split(",")
tool_name = os.environ.get("TOOL", "fastq_quality_trim_filter")
tool_args = os.environ.get("TOOL_ARGS", "-t 20 -l 30 -q 20 -p 80").split()
tool = os.path.join(ROOT, "fastx_toolkit_amd", "host", "bin", tool_name)
chunk = 250_000


def md5_of(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    return h.hexdigest()


with tempfile.TemporaryDirectory(dir="/dev/shm") as td:
    inp = os.path.join(td, "in.fq")
    t0 = time.perf_counter()
    with open(inp, "wb") as f:
        with ThreadPoolExecutor(max_workers=min(os.cpu_count() or 2, 48)) as ex:
            for part in ex.map(lambda k: fo.synth_fastq(int(os.environ.get("SEED", "2")), k * chunk, chunk, int(os.environ.get("LEN", "150")), os.environ.get("ADAPTER") == "1"), range(reads // chunk)):
                f.write(part)
    print("input: %d reads, %.2f GB, generated in %.1f s" % (reads, os.path.getsize(inp) / 1e9, time.perf_counter() - t0), flush=True)
    ref_md5 = None
    out = os.path.join(td, "out.fq")
    for item in matrix:
        env = dict(os.environ, FXH_TIMING="1")
        for kv in [x for x in item.split(":") if x]:
            env[kv.split("=")[0]] = kv.split("=", 1)[1]
        best, errtxt, walls = None, b"", []
        for rep in range(int(os.environ.get("REPS", "3"))):
            if os.path.exists(out):
                os.unlink(out)
            t0 = time.perf_counter()
            nranks = int(env.get("RANKS", "1"))
            if nranks > 1:         # a rank job on this box's ONE GPU: the ranks share it, the exchanges go through the test transport (tests/emu/fake_rccl.c, HIP copies)
                sys.path.insert(0, os.path.join(ROOT, "tests"))
                import emu_py
                renv = dict(env, LD_LIBRARY_PATH=emu_py.build_fake_rccl() + os.pathsep + env.get("LD_LIBRARY_PATH", ""), FXG_FAKE_RCCL_HIP="1", FXH_WORLD=str(nranks), FXG_DEVICE="0")
                ps = [subprocess.Popen([tool] + tool_args + ["-i", inp, "-o", out], env=dict(renv, FXH_RANK=str(r)), stderr=subprocess.PIPE) for r in range(nranks)]
                errs = [q.communicate()[1] for q in ps]
                dt = time.perf_counter() - t0
                assert all(q.returncode == 0 for q in ps), [e[-300:] for e in errs]

                class p:
                    stderr = b"".join(errs)
            else:
                p = subprocess.run([tool] + tool_args + ["-i", inp, "-o", out], env=env, stderr=subprocess.PIPE)
                dt = time.perf_counter() - t0
                assert p.returncode == 0, p.stderr[-500:]
            walls.append(dt)
            if best is None or dt < best:
                best, errtxt = dt, p.stderr
        m = md5_of(out)
        if ref_md5 is None:
            ref_md5 = m
        print("[%s] wall %s s (best %.3f)  %.1f Mreads/s  %.2f Gbases/s  out %.2f GB  md5 %s" % (
            item or "default invocation, no environment", " ".join("%.3f" % w for w in walls), best, reads / best / 1e6, reads * int(os.environ.get("LEN", "150")) / best / 1e9,
            os.path.getsize(out) / 1e9, "== one stream" if m == ref_md5 else "DIFFERS " + m), flush=True)
        for l in errtxt.decode(errors="replace").splitlines():
            if l.startswith("fxh timing") and (not l.startswith("fxh timing strand ") or os.environ.get("STRAND_LINES")):
                print("    " + l[:700], flush=True)
