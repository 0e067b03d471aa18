#!/bin/bash
# round-2 GPU call AC: HEAD re-validated -- GPU test tier, smoke, the default bench line
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02ac; mkdir -p $O
export TMPDIR=/tmp
make -s -C fastx_toolkit_amd/host 2>/dev/null
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee $O/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -1 | tee $O/smoke.txt
timeout 900 python bench.py 2>&1 | grep -v amdgpu.ids | tee $O/bench_cfg2.json | cut -c1-300
