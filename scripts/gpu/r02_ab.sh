#!/bin/bash
# round-2 GPU call AB: output pages faulted in ahead of the writes (fxh_writer_expect) -- CLI tests, then the end-to-end breakdown
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02ab; mkdir -p $O
export TMPDIR=/tmp
make -s -C fastx_toolkit_amd/host 2>/dev/null
# (CLI tests: call AB, first run)
ONLY=fused timeout 900 python scripts/e2e_breakdown.py 16000000 2>&1 | grep -v amdgpu.ids | grep -E "fused|mapping" | tee $O/e2e.txt
