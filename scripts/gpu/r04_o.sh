#!/bin/bash
# round 4, call o: roles of the clip instances (deciders / writers), opt-in via FXG_CLIP_WRITER_EVERY
mkdir -p gpurun_out/r04o
for e in 0 3 4 5 6 8; do for cfg in cfg3 cfg5; do FXG_CLIP_WRITER_EVERY=$e python scripts/clip_roles_potential.py $cfg 1 2>/dev/null | sed "s/^/E=$e /"; done; done | tee gpurun_out/r04o/roles.txt
for e in 4; do FXG_CLIP_WRITER_EVERY=$e timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "full_size_cfg3 or cfg5_pipeline or shards_reassemble" 2>&1 | tail -3; done
