#!/bin/bash
# round 5, call zz (second form): `trimmer | filter` with both sides of the pipe copying by several threads through private pipes (splice() moves the pages across
# the shared pipe), and the GPU tier's command-line tests on the same build
O=gpurun_out/r05zz; mkdir -p $O
READS=16000000 MATRIX=",FXH_NO_PIPE_FANOUT=1,FXH_PIPE_WRITERS=2:FXH_PIPE_READERS=2,FXH_PIPE_WRITERS=4:FXH_PIPE_READERS=5" timeout 600 python scripts/e2e_pipe.py > $O/e2e_pipe_b.txt 2>&1
cut -c1-330 $O/e2e_pipe_b.txt
READS=64000000 MATRIX="," timeout 600 python scripts/e2e_pipe.py > $O/e2e_pipe_64m_b.txt 2>&1
cut -c1-330 $O/e2e_pipe_64m_b.txt
timeout 900 python -m pytest tests/test_gpu_cli.py -q -m gpu > $O/pytest_gpu_cli.txt 2>&1; tail -n 3 $O/pytest_gpu_cli.txt
