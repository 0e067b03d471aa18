#!/usr/bin/env python3
"""A/B compile-time kernel variants: `build` (here, hipcc cross-compiles) then `run` (GPU box) times cfg2 with each library.

    python scripts/variants.py build            # fastx_toolkit_amd/libfxg_v_<name>.so, git-ignored, travel with gpurun
    python scripts/variants.py run              # one JSON line per variant (scripts/ablate.py in a subprocess each)
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fastx_toolkit_amd import build as _b  # noqa: E402

VARIANTS = {            # edit freely: every entry becomes fastx_toolkit_amd/libfxg_v_<name>.so
    "abl": ["-DFXG_ABLATION"],
    "abl_nonts": ["-DFXG_ABLATION", "-DFXG_V_NO_NTS"],
    "abl_ldnt": ["-DFXG_ABLATION", "-DFXG_ROWS_LD_AUX=2"],
    "abl_ldnt_nonts": ["-DFXG_ABLATION", "-DFXG_ROWS_LD_AUX=2", "-DFXG_V_NO_NTS"],
}


def lib(name):
    return os.path.join(_b.PKG, "libfxg_v_%s.so" % name)


if sys.argv[1] == "build":
    procs = [(n, subprocess.Popen([_b.hipcc()] + _b.HIPCC_FLAGS + d + [os.path.join(_b.CSRC, "fxg_engine.hip"), "-o", lib(n)])) for n, d in VARIANTS.items()]
    for n, p in procs:
        print(n, "rc", p.wait())
else:
    cfgs = os.environ.get("ABLATE") or json.dumps([["full", {}], ["decision-only", {}, False]])
    for n in (os.environ.get("VARIANTS", "").split() or VARIANTS):
        if not os.path.exists(lib(n)):
            continue
        env = dict(os.environ, FXG_LIB=lib(n), ABLATE=cfgs)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "ablate.py")], env=env, capture_output=True, text=True, timeout=300)
        for line in out.stdout.splitlines():
            print(n, line, flush=True)
        if out.returncode:
            print(n, "FAILED", out.stderr[-400:], flush=True)
