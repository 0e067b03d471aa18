#!/bin/bash
# round 5, call zz: `trimmer | filter` with the writer handing the pipe pages of its own staging region (safe with splicing readers) and the reader dealing the
# incoming pages out to three copying threads
O=gpurun_out/r05zz; mkdir -p $O
READS=16000000 MATRIX=",FXH_NO_PIPE_FANOUT=1,FXH_PIPE_READERS=2,FXH_PIPE_READERS=5,FXH_NO_VMSPLICE=1,FXH_NO_VMSPLICE=1:FXH_NO_PIPE_FANOUT=1" timeout 600 python scripts/e2e_pipe.py > $O/e2e_pipe.txt 2>&1
cut -c1-330 $O/e2e_pipe.txt
READS=64000000 MATRIX="," timeout 600 python scripts/e2e_pipe.py > $O/e2e_pipe_64m.txt 2>&1
cut -c1-330 $O/e2e_pipe_64m.txt
