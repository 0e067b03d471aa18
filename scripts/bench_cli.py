#!/usr/bin/env python3
"""End-to-end wall time of the command-line tools (file -> file on tmpfs) vs the reference driver. Not a test."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import fxoracle_py as fo  # noqa: E402

N = int(os.environ.get("READS", "2000000"))
BIN = os.path.join(ROOT, "fastx_toolkit_amd", "host", "bin")
subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "fastx_toolkit_amd", "host")])
d = "/dev/shm/fxcli"
os.makedirs(d, exist_ok=True)
inp = os.path.join(d, "in.fq")
with open(inp, "wb") as f:
    f.write(fo.synth_fastq(2, 0, N, 150))
ref = fo.ref_binary()
CASES = [
    ("fastq_quality_trimmer", ["-t", "20", "-l", "30"]),
    ("fastq_quality_filter", ["-q", "20", "-p", "80"]),
    ("fastx_trimmer", ["-f", "5", "-l", "145"]),
    ("fastx_reverse_complement", []),
    ("fastx_clipper", ["-a", "AGATCGGAAGAGC", "-l", "15", "-n"]),
]
for tool, args in CASES:
    row = [tool]
    outs = []
    for label, exe in (("gpu", [os.path.join(BIN, tool)]), ("ref", [ref, tool] if ref else None)):
        if exe is None:
            continue
        n = N if (label == "gpu" or tool != "fastx_clipper") else N
        out = os.path.join(d, "out_%s.fq" % label)
        best = 1e9
        for _ in range(2 if label == "gpu" else 1):
            t0 = time.perf_counter()
            subprocess.check_call(exe + args + ["-i", inp, "-o", out])
            best = min(best, time.perf_counter() - t0)
        outs.append(out)
        row.append("%s %.2fs %.2f Mreads/s" % (label, best, N / best / 1e6))
    if len(outs) == 2:
        same = subprocess.call(["cmp", "-s", outs[0], outs[1]]) == 0
        row.append("identical" if same else "DIFFERENT")
    print(" | ".join(row), flush=True)
