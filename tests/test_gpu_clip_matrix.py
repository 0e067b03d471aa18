"""GPU tier: clip instance x launch-bounds parity matrix (round-3 verdict item 1).

One clip instance once came out wrong on the GPU only, and only for one register budget.  The cause is a
compiler bug that moves with the register allocation: spill stores / copies of a lane-divergent loop's live-out values placed in
the loop's exit block ahead of the EXEC restore.  So every packed clip instance is built three more times -- for 2, 3 and 4 waves
per SIMD, i.e. 256, 168 and 128 registers (scripts/build_clip_matrix.py: three further allocations of the same sources) -- and

  * every instance the ISA check accepts must agree with the oracle on the whole adversarial corpus (every bucket of both packed forms,
    adapters with N, long reads) and on a seeded fuzz that names the wide instances (fixed and ragged with clip history);
  * a library in which the check names an instance is refused by the build as a whole (fastx_toolkit_amd/build.py), which is asserted;
    the named instances are never launched (round 6: one of them faulted the GPU; their cases are counted and skipped).
The shipped library is the fourth column of the matrix (tests/test_gpu_parity.py runs the same corpus through it).

Round 6: an instance is held to every budget DOWN TO the one the product builds it for (fxg_clip_waves: four waves per SIMD up to 36 columns, three up to
64, two above), not below it.  A 48-column instance squeezed into 128 registers (it ships with 168) keeps its rows in scratch, runs at half the speed --
nobody would pick it -- and ROCm 7.2 turned it into code in which ONE LANE PER WAVE writes its result (a saved EXEC mask lost under the spilling; res[] of
the other lanes keeps what the buffer held, the write-out then faults the GPU: profiles/r06/t_matrix_w4_48_columns_one_lane_per_wave.txt), a miscompile of a
class scripts/check_exec_zero.py does not recognise.  Launching such builds proves nothing about the product and can lose the box.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

WORKER = r"""
import json, os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np, torch
from helpers import adversarial_clip_cases, oracle_params, assert_same, random_batch
from oracle import fxoracle_py as fo
from fastx_toolkit_amd import Engine, make_params
eng = Engine(0)
scr = np.random.default_rng(77)
def once(b, q, lens, pd):
    dl = torch.from_numpy(np.ascontiguousarray(lens).view(np.int16)).to(eng.device) if lens is not None else None
    return eng.run(eng.upload(b).view(b.shape), eng.upload(q).view(q.shape), make_params(**pd), lens=dl, fixed_len=None if lens is not None else b.shape[1], compact=True).to_host()
def run(b, q, lens, pd):
    # A lost spill store shows only when the slot holds something ELSE: the same instance runs on unrelated reads of the same shape first,
    # so that what the waves' scratch, the checkpoint scratch and the history workspace hold is never this very launch's own earlier values
    # (that is how the four-wave 48-column instance was wrong on a first launch and right on every repetition, DESIGN.md section 3).
    b2 = np.ascontiguousarray(scr.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), size=b.shape))
    l2 = scr.integers(1, b.shape[1] + 1, size=b.shape[0]).astype(np.uint16) if lens is not None else None
    once(b2, q, l2, pd)
    if lens is not None:
        eng.set_clip_history(True)               # a fresh aligner for the real run, as for the scrambled one
    return once(b, q, lens, pd)
rejected = json.loads(sys.argv[2])                    # instances the ISA check refused in this library: NEVER launched (round 6: a refused 64-column instance at
kernels, n, refused = set(), 0, {}                    # four waves faulted the GPU -- code that runs with EXEC = 0 ahead of its restore can compute any address)
BUCKETS = [4, 8, 9, 10, 11, 12, 13, 14, 15, 16, 20, 24, 28, 32, 36, 40, 44, 48, 52, 56, 60, 64, 72, 80, 88, 100]     # fxg_plan.h
def inst_of(ad):
    if len(ad) > 16 and len(set(ad) - {ord("N")}) > 6:
        return None                                   # the general form
    return "<-%d,0>" % [b for b in BUCKETS if len(ad) <= b][0]
WAVES = int(sys.argv[3])
def shipped_waves(inst):                              # fxg_clip_waves (csrc/fxg_kernels.h): the budget the product builds the instance for
    cols = -int(inst[1:inst.index(",")])
    return 4 if cols <= 36 else 3 if cols <= 64 else 2
def skip(pd):
    inst = inst_of(pd["adapter"])
    if inst in rejected or (inst is not None and WAVES > shipped_waves(inst)):
        refused[inst] = refused.get(inst, 0) + 1      # refused by the ISA check, or a budget with FEWER registers than the product's (module docstring)
        return True
    return False
def compare(o, e, name):
    global n
    k = eng.last_launch()["kernel"].split()[0]
    inst = k[k.index("<"):]
    assert inst not in rejected, (inst, name)         # the table above and the plan must agree on who is refused
    assert_same(o, e, name)
    kernels.add(k); n += 1
    return k
for long_adapters in (False, True):
    for name, b, q, pd in adversarial_clip_cases(long_adapters):
        if skip(pd):
            continue
        compare(fo.run_pipeline(b, q, None, oracle_params(pd)), run(b, q, None, pd), name)
# the wide instances by name (48 .. 100 columns, with and without N in the adapter: an N is a column pattern of the pair table, the instance is the same), across strides, fixed and ragged with clip history
rng = np.random.default_rng(404)
ADS = {"-44": b"ACGTTGCAAGGCTTAACCGGATATCGCGTATAGCTAGCTAGGATCCA"[:44], "-64": b"GTCGTAGACCGATCGGGGACCCCTTGTTTCACGCGTCGTATAGCTGCTATGTCATTAGCAAGG"[:62],
       "-56": b"GTCGTAGACCGATCGGGGACCCCTTGTTTCACGCGTCGTATAGCTGCTATGTCATTAGC"[:53], "-80": (b"ACGTTGCAAGGCTTAACCGGATATCGCGTATAG" * 3)[:77],
       "-52 N": b"AGATCGGAAGAGCACACGTCTGAACTCCAGTCACNNNNNNATCTCGTATGCCGTCTTC"[:51], "-72 N": b"AGATCGGAAGAGCACACGTCTGAACTCCAGTCACNNNNNNATCTCGTATGCCGTCTTCTGCTTGAAAAAAAAAAGGGGGG"[:72],
       "-100": (b"ACGTTGCA" * 12)[:91], "-48 N": b"AGATCGGAAGAGCACACGTCTGAACTCCAGTCACNNNNNNATCTCG"[:46], "-60 N": b"AGATCGGAAGAGCACACGTCTGAACTCCAGTCACNNNNNNATCTCGTATGCCGTCTTC"[:58], "-88": (b"ACGTTGCAAGGCTTAACCGGATATCGCGTATAG" * 3)[:85],
       "-100 N": b"AGATCGGAAGAGCACACGTCTGAACTCCAGTCACNNNNNNATCTCGTATGCCGTCTTCTGCTTGAAAAAAAAAAGGGGGGGGGGCCCCCCCCCCTTTTT"[:93]}
for tag, ad in ADS.items():
    tag = tag.split(" ")[0]
    for stride in (150, 151, 250, 300):
        for fixed in (True, False):
            nreads = int(rng.integers(200, 700))
            b, q, lens = random_batch(rng, nreads, stride, max(1, stride // 3), stride, fixed, adapter=ad)
            pd = dict(stages=1, adapter=ad, clip_min_len=int(rng.integers(0, 20)), clip_flags=int(rng.integers(0, 16)), clip_min_adapter_len=int(rng.choice([0, 0, 5])))
            if skip(pd):
                continue
            eng.set_clip_history(not fixed)
            k = compare(fo.run_pipeline(b, q, lens, oracle_params(pd)), run(b, q, lens, pd), "wide%s.s%d.%s" % (tag, stride, "fixed" if fixed else "ragged+history"))
            eng.set_clip_history(False)
            assert ("<%s," % tag) in k, (tag, k)
print(json.dumps(dict(cases=n, kernels=sorted(kernels), refused=refused)))
"""


@pytest.fixture(scope="module")
def matrix():
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import build_clip_matrix as m
    return m, m.build_all()          # built here beforehand (the .so files travel); rebuilt on the box only if stale


@pytest.mark.parametrize("waves", [2, 3, 4])
def test_clip_instances_at_every_register_budget(matrix, waves):
    m, verdicts = matrix
    v = verdicts[waves]
    from fastx_toolkit_amd import build as b
    if not v["accepted"]:                 # the build refuses this library as a whole ...
        with pytest.raises(RuntimeError, match="REJECTED"):
            b.check_exec_zero(m.lib(waves))
    # ... the matrix still runs it: every instance the check did NOT name must agree with the oracle; the named ones are not launched
    # (nobody ships them, and code that runs ahead of its EXEC restore can compute any address: one faulted the GPU in round 6)
    p = subprocess.run([sys.executable, "-c", WORKER, ROOT, json.dumps(v["rejected_instances"]), str(waves)], env=dict(os.environ, FXG_LIB=m.lib(waves)),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1500)
    assert p.returncode == 0, "waves %d: an instance the ISA check accepted disagrees with the oracle:\n%s" % (waves, p.stderr[-3000:])
    d = json.loads(p.stdout.strip().splitlines()[-1])
    print("waves %d: %d cases equal to the oracle; refused instances (cases not launched): %s" % (waves, d["cases"], d["refused"]))
    ran = {k[k.index("<"):] for k in d["kernels"]} | set(v["rejected_instances"])
    for inst in ("-4", "-8", "-13", "-16", "-20", "-24", "-28", "-32", "-36", "-40", "-44", "-48", "-52", "-56", "-60", "-64", "-72", "-80", "-88", "-100"):
        cols = -int(inst)
        if waves <= (4 if cols <= 36 else 3 if cols <= 64 else 2):      # (an instance is held to every budget down to the one it ships with)
            assert "<%s,0>" % inst in ran, (inst, sorted(ran))
    assert d["cases"] > (400 if waves == 2 else 250 if waves == 3 else 150)
