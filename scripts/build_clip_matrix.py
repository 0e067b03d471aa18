#!/usr/bin/env python3
"""Builds the libraries of the clip instance x launch-bounds parity matrix (tests/test_gpu_clip_matrix.py): the whole engine with EVERY
packed clip instance compiled for w waves per SIMD, w in {2, 3, 4} -- i.e. for 256, 168 and 128 registers: three different register
allocations of every instance.  Each library goes through the ISA check of scripts/check_exec_zero.py; its verdict is written next to
it (libfxg_m_w<N>.json) and a rejected library is kept (as evidence, and so that the test can show that a rejected build is refused),
but the test never expects right answers from the instances the check names (rejected_instances); the others it still runs.

    python scripts/build_clip_matrix.py            # here (hipcc cross-compiles; the .so files travel to the GPU box, git ignores them)
"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fastx_toolkit_amd import build as b  # noqa: E402

WAVES = (2, 3, 4)


def lib(w):
    return os.path.join(b.PKG, "libfxg_m_w%d.so" % w)


def stale(w):
    deps = [os.path.join(b.CSRC, f) for f in os.listdir(b.CSRC)] + [os.path.join(ROOT, "include", "fxg.h")]
    return b._newer(lib(w), deps)


def verdict(w):
    """The ISA check's answer for one library (always computed afresh: the checker may have learnt since the library was built)."""
    chk = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_exec_zero.py"), lib(w)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    rejected = sorted({"<%s,%s>" % (a.replace("n", "-"), m) for a, m in re.findall(r"fxg_kernel_tilesILi(n?\d+)ELi(\d+)E(?:Lb[01]E)?Ev", chk.stdout)})
    v = dict(waves=w, accepted=chk.returncode == 0, rejected_instances=rejected, report=chk.stdout[-4000:])
    json.dump(v, open(lib(w)[:-3] + ".json", "w"), indent=1)
    return v


def build_all(waves=WAVES):
    procs = [(w, subprocess.Popen([b.hipcc()] + b.HIPCC_FLAGS + ["-DFXG_CLIP_WAVES_ALL=%d" % w, os.path.join(b.CSRC, "fxg_engine.hip"), "-o", lib(w)]))
             for w in waves if stale(w)]
    for w, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed for the %d-wave matrix library" % w)
    return {w: verdict(w) for w in waves}


if __name__ == "__main__":
    for w, v in build_all().items():
        print("waves %d: %s" % (w, "accepted" if v["accepted"] else "REJECTED by the ISA check\n" + v["report"]))
