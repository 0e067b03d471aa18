#!/bin/bash
# round-2 GPU call F: extended device text path (CRLF, numeric qualities, FASTA) parity; end-to-end breakdown with the mapped writer
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02f; mkdir -p $O
export TMPDIR=/tmp
make -s -C fastx_toolkit_amd/host 2>/dev/null
echo "== text path tests"
timeout 900 python -m pytest tests/test_gpu_text.py tests/test_gpu_cli.py -m gpu -q 2>&1 | tail -25 | tee $O/pytest.txt
echo "== e2e breakdown"
timeout 600 python scripts/e2e_breakdown.py 16000000 2>&1 | grep -v amdgpu.ids | tee $O/e2e_breakdown.txt
