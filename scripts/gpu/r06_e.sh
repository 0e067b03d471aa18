#!/bin/bash
# round 6, call e: pass 2 with wave-uniform loops again; SQ counters; the tuples of every rank's shard (bench.EXPECTED)
O=gpurun_out/r06e; mkdir -p $O
LIBS=fastx_toolkit_amd/libfxg_v_f32.so,fastx_toolkit_amd/libfxg.so timeout 900 python scripts/clip_ab.py > $O/clip_ab.txt 2>&1
cut -c1-330 $O/clip_ab.txt
FXG_LIB=$PWD/fastx_toolkit_amd/libfxg.so CFG=cfg3 bash scripts/pmc_sq.sh r06e/sq_cfg3_head scripts/pmc_clip.py 2>&1 | grep -E "INSTS_VALU|INSTS_LDS|WAVE_CYCLES|BANK_CONFLICT|WAIT_INST_LDS|ACTIVE_INST_VALU|ACTIVE_INST_ANY|WAIT_ANY" | grep tiles
timeout 1500 python scripts/pin_shards.py > $O/pin_shards.txt 2>&1; tail -n 20 $O/pin_shards.txt
