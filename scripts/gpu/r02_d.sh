#!/bin/bash
# round-2 GPU call D: full GPU test tier, clip occupancy A/B, cfg4 tile sweep, all bench configs, end-to-end tool timing
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02d; mkdir -p $O
export TMPDIR=/tmp
echo "== gpu tests"
timeout 1200 python -m pytest tests -m gpu -q -x --durations=8 2>&1 | tail -25 | tee $O/pytest.txt
echo "== clip A/B (waves per SIMD)"
( ONLY=cfg3 timeout 200 python scripts/bench_configs.py; FXG_LIB=fastx_toolkit_amd/libfxg_v_clipw1.so ONLY=cfg3 timeout 200 python scripts/bench_configs.py; ONLY=cfg5 timeout 200 python scripts/bench_configs.py ) 2>&1 | grep -v amdgpu.ids | tee $O/clip.txt
echo "== cfg4 tile sweep"
CFG=cfg4 ABLATE='[["t256",{"FXG_TILE":"256"}],["t128",{"FXG_TILE":"128"}],["t64",{"FXG_TILE":"64"}]]' FXG_LIB=fastx_toolkit_amd/libfxg.so timeout 300 python scripts/ablate.py 2>&1 | grep -v amdgpu.ids | tee $O/cfg4.txt
echo "== cfg2 tile sweep"
CFG=cfg2 ABLATE='[["t256",{"FXG_TILE":"256"}],["t128",{"FXG_TILE":"128"}],["decision-only",{},false]]' FXG_LIB=fastx_toolkit_amd/libfxg.so timeout 300 python scripts/ablate.py 2>&1 | grep -v amdgpu.ids | tee $O/cfg2.txt
echo "== bench configs"
for c in cfg2 cfg3 cfg4 cfg5shard stats; do timeout 400 python bench.py --config $c --steps 10 --warmup 2 2>&1 | grep -v amdgpu.ids | tee $O/bench_$c.json | cut -c1-1500; done
echo "== e2e"
timeout 900 python scripts/e2e_cli.py 16000000 2>&1 | grep -v amdgpu.ids | tee $O/e2e.txt
