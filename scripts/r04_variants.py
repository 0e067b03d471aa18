#!/usr/bin/env python3
"""Round-4 kernel experiments: `build` cross-compiles variant libraries here, `run` times cfg3 / cfg5 (20 M reads) and the statistics kernel (50 M) with each on the GPU box."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fastx_toolkit_amd import build as b
VARIANTS = {
    "base": [],
    "ticket_after_dp": ["-DFXG_TICKET_AFTER_DP=1"],
    "ticket_last_moment": ["-DFXG_TICKET_AFTER_DP=2"],
    "qs_pipe": [],
    "qs_fake_banks": ["-DFXG_QS_FAKE_BANKS"],
    "qs_pipe_u1": ["-DFXG_QS_UNROLL=1u"],
    "qs_pipe_u3": ["-DFXG_QS_UNROLL=3u"],
    "qs_unroll3": ["-DFXG_QS_UNROLL=3u"],
    "qs_unroll4": ["-DFXG_QS_UNROLL=4u"],
    "qs_unroll1": ["-DFXG_QS_UNROLL=1u"],
    "abl": ["-DFXG_ABLATION"],
}
def lib(n): return os.path.join(b.PKG, "libfxg_x_%s.so" % n)
if sys.argv[1] == "build":
    names = sys.argv[2:] or list(VARIANTS)
    procs = [(n, subprocess.Popen([b.hipcc()] + b.HIPCC_FLAGS + VARIANTS[n] + [os.path.join(b.CSRC, "fxg_engine.hip"), "-o", lib(n)])) for n in names]
    for n, p in procs:
        rc = p.wait()
        chk = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_exec_zero.py"), lib(n)], capture_output=True, text=True)
        print(n, "rc", rc, "isa check:", chk.stdout.strip().splitlines()[-1] if chk.stdout.strip() else chk.stderr[-200:], flush=True)
else:
    names = sys.argv[2:] or [n for n in VARIANTS if os.path.exists(lib(n))]
    for n in names:
        env = dict(os.environ, FXG_LIB=lib(n))
        if n.startswith("qs_") or n == "base":
            p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "stats", "--no-cpu-baseline", "--steps", "10"], env=env, capture_output=True, text=True)
            try:
                d = json.loads(p.stdout.strip().splitlines()[-1]); print(n, "stats", d["roofline"]["kernel_ms_avg"], d["roofline"]["frac"], flush=True)
            except Exception as e:
                print(n, "stats ERR", p.stderr[-300:])
        if not n.startswith("qs_"):
            for cfg in ("cfg3", "cfg5"):
                for extra in ({}, {"FXG_TILE": "128"}):
                    p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "clip_roles_potential.py"), cfg, "1"], env=dict(env, **extra), capture_output=True, text=True)
                    print(n, extra, p.stdout.strip() or p.stderr[-300:], flush=True)
