#!/usr/bin/env python3
"""What the alignment of the write-out's source windows is worth (GPU box, not a test): the fixed trimmer with a no-op range (source window of an output chunk = the
chunk's own offset: aligned 16-byte loads) against `-f 5 -l 145` (misaligned), the reverse complement on rows of 160 bytes (reversed windows aligned) against 150."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fastx_toolkit_amd import Engine, make_params  # noqa: E402

eng = Engine(0)
eng.set_profiling(True)
CASES2 = [(144, "fastx_reverse_complement, 144-byte rows (aligned, no chunk across reads)", dict(stages=8)), (152, "fastx_reverse_complement, 152-byte rows (windows at 8 mod 16, every other read straddled)", dict(stages=8)),
          (158, "fastx_reverse_complement, 158-byte rows", dict(stages=8)), (150, "fastx_reverse_complement, 150-byte rows", dict(stages=8)), (150, "fastx_trimmer -f 3 -l 150 (shift of 2)", dict(stages=16, ft_first=3, ft_last=150)),
          (150, "fastx_trimmer -t 2 (150 -> 148, end trimmed)", dict(stages=32, ft_trim_end=2)), (150, "revcomp + trimmer -f 1 -l 150", dict(stages=24, ft_first=1, ft_last=150))]
CASES3 = [(150, "revcomp + trimmer -f 1 -l %d (window start mod 4: %s)" % (l, w), dict(stages=24, ft_first=1, ft_last=l)) for l, w in ((150, "2"), (148, "0 / 2 by read"), (147, "all"), (146, "2"), (144, "0"))]
for L, name, pd in CASES3 if os.environ.get("CASES") == "3" else CASES2 if os.environ.get("CASES") == "2" else [(150, "fastx_trimmer -f 1 -l 150 (no-op, aligned windows)", dict(stages=16, ft_first=1, ft_last=150)), (150, "fastx_trimmer -f 5 -l 145", dict(stages=16, ft_first=5, ft_last=145)),
                    (150, "fastx_trimmer -f 17 -l 150 (shift of 16: aligned)", dict(stages=16, ft_first=17, ft_last=150)), (150, "fastx_trimmer -f 2 -l 150 (shift of 1)", dict(stages=16, ft_first=2, ft_last=150)),
                    (150, "fastx_reverse_complement, 150-byte rows", dict(stages=8)), (160, "fastx_reverse_complement, 160-byte rows (aligned windows)", dict(stages=8)),
                    (150, "fastq_masker -q 20, 150-byte rows", dict(stages=64, mask_min_quality=20)), (160, "fastx_trimmer -f 5 -l 145, 160-byte rows", dict(stages=16, ft_first=5, ft_last=145))]:
    N = 50_000_000 * 150 // L
    b, q = eng.synth(2, 0, N, L, False)
    outs = eng.alloc_outputs(N, L, compact=True, meta=False)
    P = make_params(**pd)
    ms = []
    for _ in range(5):
        r = eng.run(b, q, P, fixed_len=L, compact=True, meta=False, outputs=outs)
        ms.append(eng.last_kernel_ms())
    kb = int(r.counters[2])
    alg = N * (2 * L + 4) + 2 * kb
    t = min(ms) * 1e-3
    print(json.dumps(dict(case=name, kernel=eng.last_launch()["kernel"], ms_min=round(min(ms), 3), alg_GB=round(alg / 1e9, 2), alg_TBs=round(alg / t / 1e12, 2), frac_hbm=round(alg / t / 8e12, 3))), flush=True)
    del b, q, outs, r
    import torch
    torch.cuda.empty_cache()
