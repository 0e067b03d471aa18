#!/usr/bin/env python3
"""Start / end-of-loop / end clocks of every workgroup of the statistics kernel (GPU box, a -DFXG_QS_CLOCKS build as FXG_LIB; not a test):
how evenly the workgroups of the persistent launch finish."""
import collections, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fastx_toolkit_amd import Engine
eng = Engine(0)
n, L = int(os.environ.get("READS", "50000000")), 150
b, q = eng.synth(2, 0, n, L, False)
eng.set_profiling(True)
for rep in range(4):
    h = eng.quality_stats(b, q, fixed_len=L, sync=True).cpu().numpy()
    ms = eng.last_kernel_ms()
    G = eng.last_launch()["grid"]
    c = [[int(h[g // 5, g % 5, k]) for k in range(3)] for g in range(G)]
    t0 = min(x[0] for x in c)
    st, e1, e2 = (sorted((x[k] - t0) / 100.0 for x in c) for k in range(3))
    qq = lambda v, f: v[int(f * (len(v) - 1))]
    print("launch %d (%.3f ms): start us max %.1f | loop end us min/p10/med/p90/max %.1f %.1f %.1f %.1f %.1f | end min/max %.1f %.1f" %
          (rep, ms, st[-1], qq(e1, 0), qq(e1, .1), qq(e1, .5), qq(e1, .9), qq(e1, 1), e2[0], e2[-1]))
byx = collections.defaultdict(list)
for g, x in enumerate(c):
    byx[g % 8].append((x[1] - t0) / 100.0)
print("last launch, loop end by blockIdx % 8: " + "  ".join("%d: %.0f..%.0f (mean %.0f)" % (x, min(v), max(v), sum(v) / len(v)) for x, v in sorted(byx.items())))
