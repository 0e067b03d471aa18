#!/bin/bash
# phase timing of the CLI host path at 8M reads (GPU box)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
make -s -C fastx_toolkit_amd/host 2>/dev/null
python -c "
import sys; sys.path.insert(0,'.')
from oracle import fxoracle_py as fo
open('/dev/shm/in.fq','wb').write(fo.synth_fastq(2,0,${READS:-8000000},150))"
for t in "fastq_quality_trimmer -t 20 -l 30" "fastx_reverse_complement" "fastx_clipper -a AGATCGGAAGAGC -l 15 -n"; do
  s=$(date +%s.%N)
  FXH_TIMING=1 fastx_toolkit_amd/host/bin/$t -i /dev/shm/in.fq -o /dev/shm/out.fq 2>&1 | grep -v amdgpu
  e=$(date +%s.%N)
  python3 -c "print(\"$t : wall %.2f s\" % ($e - $s))"
done
# -z: parallel gzip members vs what the reference does (pipe through one gzip process)
s=$(date +%s.%N); fastx_toolkit_amd/host/bin/fastq_quality_trimmer -t 20 -l 30 -z -i /dev/shm/in.fq -o /dev/shm/out.fq.gz 2>&1 | grep -v amdgpu; e=$(date +%s.%N)
python3 -c "print(\"fastq_quality_trimmer -z (parallel deflate): wall %.2f s\" % ($e - $s))"
if [ "${WITH_SINGLE_GZIP:-0}" = "1" ]; then   # two minutes of one gzip process: measured once (122 s vs 6.7 s), off by default
s=$(date +%s.%N); fastx_toolkit_amd/host/bin/fastq_quality_trimmer -t 20 -l 30 -i /dev/shm/in.fq | gzip > /dev/shm/out2.fq.gz; e=$(date +%s.%N)
python3 -c "print(\"fastq_quality_trimmer | gzip (one gzip process, as the reference's -z): wall %.2f s\" % ($e - $s))"
zcat /dev/shm/out.fq.gz | md5sum; zcat /dev/shm/out2.fq.gz | md5sum; ls -l /dev/shm/out.fq.gz /dev/shm/out2.fq.gz | awk '{print $5, $9}'
fi
