#!/bin/bash
# round 6, call d: SQ counters of the clip kernel (cfg3, 20 M reads): compare + select cell / float table in pass 1 / float table in both passes
for v in noptab f32 head; do
  lib=fastx_toolkit_amd/libfxg_v_$v.so; [ $v = head ] && lib=fastx_toolkit_amd/libfxg.so
  echo "== $v"
  FXG_LIB=$PWD/$lib CFG=cfg3 bash scripts/pmc_sq.sh r06d/sq_cfg3_$v scripts/pmc_clip.py 2>&1 | tail -12
done
