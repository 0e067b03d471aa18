#!/usr/bin/env python3
"""Bisect of the 64-column clip instance that came out wrong on the GPU when compiled for three waves per SIMD.

    python scripts/debug/clip64_bisect.py build      # here: cross-compiles the variant libraries (fastx_toolkit_amd/libfxg_v_<name>.so)
    python scripts/debug/clip64_bisect.py run        # GPU box: every variant x the adversarial cases of the wide buckets, against the oracle
    python scripts/debug/clip64_bisect.py emu        # here: the same dumps from the CPU emulator (truth for every stage)
    python scripts/debug/clip64_bisect.py analyze    # here, afterwards: first stage of every read that differs from the emulator's dump
    python scripts/debug/clip64_bisect.py one NAME   # (internal) one variant in its own process

`run` prints, per variant, the cases that differ from the oracle with the failing reads' lane / wave, the position of the planted
adapter (the oracle's clip point) and -- for the -DFXG_CLIP_DEBUG builds -- the first stage of fxg_clip_two_pass_k whose dumped value
differs from the same dump of the two-wave build (pass 1: bq1 / b1; restart: r0, j0, hash of the checkpoint row; re-run: hash of S
at r0; summary rows: hashes of S and W before the last row; last row: best, bw).
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fastx_toolkit_amd import build as _b  # noqa: E402

W3 = ["-DFXG_CLIP_WAVES_WIDE=3"]
D2, D3 = ["-DFXG_CLIP_DEBUG=2"], ["-DFXG_CLIP_DEBUG=3"]
W4 = ["-DFXG_CLIP_WAVES_ALL=4"]
VARIANTS_W4 = {                                # CASES=wide-348: the 48-column N instance at four waves per SIMD (wrong on ragged input with clip history, call r04_c)
    "w4": (W4, "-O3"),
    "w4_dbg1": (W4 + ["-DFXG_CLIP_DEBUG=1"], "-O3"),
    "w4_dbg2": (W4 + ["-DFXG_CLIP_DEBUG=2"], "-O3"),
    "w4_dbg3": (W4 + ["-DFXG_CLIP_DEBUG=3"], "-O3"),
    "w3all": (["-DFXG_CLIP_WAVES_ALL=3"], "-O3"),
    "w3all_dbg3": (["-DFXG_CLIP_WAVES_ALL=3", "-DFXG_CLIP_DEBUG=3"], "-O3"),
}
VARIANTS = {                                  # name: (extra flags, optimisation level)
    "w2_dbg2": (D2, "-O3"),                                           # wrong on the GPU in calls r04_a, r04_b (the shipped allocation + hash dumps)
    "w2_dbg2_W0": (D2 + ["-DFXG_CLIP_DEBUG_W=0"], "-O3"),             # + all of W before the last row
    "w2_dbg2_W40": (D2 + ["-DFXG_CLIP_DEBUG_W=40"], "-O3"),           # + W[40..63]
    "w2_dbg2_W46": (D2 + ["-DFXG_CLIP_DEBUG_W=46"], "-O3"),
    "w2_dbg1_W0": (["-DFXG_CLIP_DEBUG=1", "-DFXG_CLIP_DEBUG_W=0"], "-O3"),
    "w2_dbg1_W40": (["-DFXG_CLIP_DEBUG=1", "-DFXG_CLIP_DEBUG_W=40"], "-O3"),
    "w2_dbg2_S0": (D2 + ["-DFXG_CLIP_DEBUG_S=0"], "-O3"),
    "w3_sgprmem": (W3 + ["-mllvm", "-amdgpu-spill-sgpr-to-vgpr=0"], "-O3"),       # wrong in r04_a, r04_b (no instrumentation at all)
    "w3_sgprmem_dbg1_W0": (W3 + ["-DFXG_CLIP_DEBUG=1", "-DFXG_CLIP_DEBUG_W=0", "-mllvm", "-amdgpu-spill-sgpr-to-vgpr=0"], "-O3"),
    "w3_sgprmem_dbg2": (W3 + D2 + ["-mllvm", "-amdgpu-spill-sgpr-to-vgpr=0"], "-O3"),
}
STAGES = ["bq1", "b1", "r0", "j0", "ck_row_hash", "S_at_r0_hash", "best", "bw", "bq", "-", "S_before_last_hash", "W_before_last_hash",
          "S_after_last_hash", "W_after_last_hash", "-", "-"]


def lib(name):
    return os.path.join(_b.PKG, "libfxg_v_%s.so" % name)


def build(names):
    procs = []
    for n in names:
        extra, opt = VARIANTS[n]
        flags = [f for f in _b.HIPCC_FLAGS if f != "-O3"] + [opt]
        procs.append((n, subprocess.Popen([_b.hipcc()] + flags + extra + [os.path.join(_b.CSRC, "fxg_engine.hip"), "-o", lib(n)])))
    for n, p in procs:
        print(n, "rc", p.wait(), flush=True)


WIDE_ADS = {"-48": b"ACGTTGCAAGGCTTAACCGGATATCGCGTATAGCTAGCTAGGATCCA"[:44], "-64": b"GTCGTAGACCGATCGGGGACCCCTTGTTTCACGCGTCGTATAGCTGCTATGTCATTAGC"[:57],
            "-100": (b"ACGTTGCA" * 12)[:91], "-348": b"AGATCGGAAGAGCACACGTCTGAACTCCAGTCACNNNNNNATCTCG"[:46],
            "-364": b"AGATCGGAAGAGCACACGTCTGAACTCCAGTCACNNNNNNATCTCGTATGCCGTCTTC"[:58],
            "-400": b"AGATCGGAAGAGCACACGTCTGAACTCCAGTCACNNNNNNATCTCGTATGCCGTCTTCTGCTTGAAAAAAAAAAGGGGGG"[:80]}


def wide_cases(only=None):
    """The seeded fuzz of tests/test_gpu_clip_matrix.py that names the wide instances: (name, bases, qual, lens, params, history)."""
    import numpy as np
    from helpers import random_batch
    rng = np.random.default_rng(404)
    for tag, ad in WIDE_ADS.items():
        for stride in (150, 151, 250, 300):
            for fixed in (True, False):
                nreads = int(rng.integers(200, 700))
                b, q, lens = random_batch(rng, nreads, stride, max(1, stride // 3), stride, fixed, adapter=ad)
                pd = dict(stages=1, adapter=ad, clip_min_len=int(rng.integers(0, 20)), clip_flags=int(rng.integers(0, 16)), clip_min_adapter_len=int(rng.choice([0, 0, 5])))
                if only is None or tag in only:
                    yield "wide%s.s%d.%s" % (tag, stride, "fixed" if fixed else "ragged+history"), b, q, lens, pd, not fixed


def cases():
    """(name, bases, qual, lens, params, clip history on) -- CASES=wide-348,... selects the wide fuzz of those instances instead of the adversarial corpus"""
    sel = os.environ.get("CASES", "")
    if sel.startswith("wide"):
        yield from wide_cases(sel[4:].split(",") if len(sel) > 4 else None)
        return
    from helpers import adversarial_clip_cases
    for name, b, q, pd in adversarial_clip_cases(True):
        if len(pd["adapter"]) >= 49:
            yield name, b, q, None, pd, False


OUT = os.path.join(ROOT, "gpurun_out", os.environ.get("BISECT_OUT", "r04_bisect"))
DBG_WORDS = 512


def one(name):
    import numpy as np
    from fastx_toolkit_amd import Engine, make_params
    dump = "/tmp/clipdbg_%s.bin" % name
    if "dbg" in name:
        os.environ["FXG_CLIP_DEBUG_OUT"] = dump
    eng = Engine(0)
    out = {}
    import torch
    for k, (cname, b, q, lens, pd, hist) in enumerate(cases()):
        if os.path.exists(dump):
            os.remove(dump)
        dl = torch.from_numpy(np.ascontiguousarray(lens).view(np.int16)).to(eng.device) if lens is not None else None
        eng.set_clip_history(hist)
        r = eng.run(eng.upload(b).view(b.shape), eng.upload(q).view(q.shape), make_params(**pd), lens=dl, fixed_len=None if lens is not None else b.shape[1], compact=True).to_host()
        eng.set_clip_history(False)
        out["res%d" % k] = r["res"]
        if os.path.exists(dump):
            out["dump%d" % k] = np.fromfile(dump, dtype=np.uint32).reshape(-1, DBG_WORDS)[-b.shape[0]:]
    out["kernel"] = np.array(eng.last_launch()["kernel"].split()[0])
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, "%s.npz" % name), **out)


def emu():
    """The same dumps from the CPU emulator (the kernels' own per-thread code, serially): the truth for every stage."""
    import ctypes as C
    import numpy as np
    os.environ["FXG_EMU_DEFS"] = "-DFXG_CLIP_DEBUG=3"
    import emu_py
    from oracle import fxoracle_py as fo
    L = emu_py.lib()
    L.fxg_emu_clip_debug.restype = C.c_size_t
    L.fxg_emu_clip_debug.argtypes = [C.c_void_p, C.c_size_t]
    out = {}
    for k, (cname, b, q, lens, pd, hist) in enumerate(cases()):
        h = emu_py.hist_new() if hist else None
        out["res%d" % k] = emu_py.run_pipeline(b, q, lens, fo.make_params(**pd), hist=h)["res"]
        if h:
            emu_py.hist_free(h)
        d = np.zeros(b.shape[0] * DBG_WORDS, dtype=np.uint32)
        L.fxg_emu_clip_debug(d.ctypes.data, d.size)
        out["dump%d" % k] = d.reshape(-1, DBG_WORDS)
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, "emu.npz"), **out)


def run(names):
    import numpy as np
    from helpers import oracle_params
    from oracle import fxoracle_py as fo
    want = [(cname, fo.run_pipeline(b, q, lens, oracle_params(pd))["res"], b, pd) for cname, b, q, lens, pd, hist in cases()]
    for n in names:
        if not os.path.exists(lib(n)):
            print(n, "not built")
            continue
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "one", n], env=dict(os.environ, FXG_LIB=lib(n)), capture_output=True, text=True, timeout=600)
        if p.returncode:
            print(n, "FAILED", p.stderr[-600:])
            continue
        g = np.load(os.path.join(OUT, "%s.npz" % n))
        bad = [(cname, int((g["res%d" % k] != ores).sum()), len(pd["adapter"]), b.shape) for k, (cname, ores, b, pd) in enumerate(want) if (g["res%d" % k] != ores).any()]
        print("%-18s %s cases that differ from the oracle: %d of %d  %s" % (n, g["kernel"], len(bad), len(want), [(c[:26], k, a, sh) for c, k, a, sh in bad]), flush=True)


def analyze(names):
    """Here, after the GPU call: every dumped stage of every variant against the emulator's dump, in the order the stages are produced."""
    import numpy as np
    from helpers import oracle_params
    from oracle import fxoracle_py as fo
    t = np.load(os.path.join(OUT, "emu.npz"))
    order = [(0, "bq1"), (1, "b1"), (2, "r0"), (3, "j0"), (4, "ck_row_hash"), (5, "S_at_r0_hash")] + [(16 + k, "row%d.%s" % (k // 2, "SW"[k % 2])) for k in range(256)] + \
            [(10, "S_before_last_hash"), (11, "W_before_last_hash"), (6, "best"), (7, "bw"), (8, "bq"), (12, "S_after_last_hash"), (13, "W_after_last_hash")]
    for n in names:
        f = os.path.join(OUT, "%s.npz" % n)
        if not os.path.exists(f):
            continue
        g = np.load(f)
        for k, (cname, b, q, lens, pd, hist) in enumerate(cases()):
            ores = fo.run_pipeline(b, q, lens, oracle_params(pd))["res"]
            wrong = np.nonzero(g["res%d" % k] != ores)[0]
            if "dump%d" % k not in g:
                if len(wrong):
                    print("%-18s %s A=%d L=%d n=%d: %d reads wrong (no dump): %s" % (n, cname[:26], len(pd["adapter"]), b.shape[1], b.shape[0], len(wrong), wrong[:12]))
                continue
            dv, tv = g["dump%d" % k], t["dump%d" % k]
            firsts = {}
            for i in range(b.shape[0]):
                for w, label in order:
                    if tv[i, w] != 0xEEEEEEEE and dv[i, w] != 0xEEEEEEEE and dv[i, w] != tv[i, w]:
                        firsts.setdefault(label, []).append(i)
                        break
            if firsts or len(wrong):
                print("%-18s %s A=%d L=%d n=%d: %d reads wrong; first differing stage -> reads: %s" % (
                    n, cname[:26], len(pd["adapter"]), b.shape[1], b.shape[0], len(wrong), {l: (len(v), v[:6]) for l, v in firsts.items()}))


if __name__ == "__main__":
    VARIANTS.update(VARIANTS_W4)
    names = sys.argv[2:] or [n for n in VARIANTS if n not in VARIANTS_W4]
    if sys.argv[1] == "build":
        build(names)
    elif sys.argv[1] == "one":
        one(sys.argv[2])
    elif sys.argv[1] == "emu":
        emu()
    elif sys.argv[1] == "analyze":
        analyze(names)
    else:
        run(names)
