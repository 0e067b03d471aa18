#!/bin/bash
# round-2 GPU call E: full GPU test tier again; where the end-to-end wall time of the tools goes
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02e; mkdir -p $O
export TMPDIR=/tmp
make -s -C fastx_toolkit_amd/host 2>/dev/null
echo "== e2e breakdown"
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $O/e2e_breakdown.txt
import os, subprocess, sys, time
sys.path.insert(0, '.')
from concurrent.futures import ThreadPoolExecutor
from oracle import fxoracle_py as fo
R = 16_000_000; chunk = 250_000
with ThreadPoolExecutor(max_workers=32) as ex:
    parts = list(ex.map(lambda k: fo.synth_fastq(2, k * chunk, chunk, 150, False), range(R // chunk)))
with open('/dev/shm/in.fq', 'wb') as f:
    for p in parts: f.write(p)
with open('/dev/shm/tiny.fq', 'wb') as f: f.write(parts[0][:320000])
del parts
B = 'fastx_toolkit_amd/host/bin/'
def run(label, argv, env=None):
    e = dict(os.environ, FXH_TIMING='1', **(env or {}))
    best = None
    for _ in range(2):
        t0 = time.perf_counter(); p = subprocess.run(argv, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e); dt = time.perf_counter() - t0
        if best is None or dt < best[0]: best = (dt, p.stderr.decode(errors='replace'))
    tl = ' | '.join(l for l in best[1].splitlines() if l.startswith('fxh timing'))
    print('%-42s wall %.3f s  %s' % (label, best[0], tl[tl.find('run'):] if 'run' in tl else tl), flush=True)
T = [B + 'fastq_quality_trimmer', '-t', '20', '-l', '30', '-i', '/dev/shm/in.fq']
run('tiny input (1000 reads): startup', [B + 'fastq_quality_trimmer', '-t', '20', '-l', '30', '-i', '/dev/shm/tiny.fq', '-o', '/dev/shm/out.fq'])
run('lanes=1 -> tmpfs file', T + ['-o', '/dev/shm/out.fq'], {'FXH_LANES': '1'})
run('lanes=1 -> /dev/null', T + ['-o', '/dev/null'], {'FXH_LANES': '1'})
run('lanes=2 -> /dev/null', T + ['-o', '/dev/null'], {'FXH_LANES': '2'})
run('lanes=2 -> tmpfs file', T + ['-o', '/dev/shm/out.fq'], {'FXH_LANES': '2'})
run('lanes=2 -> tmpfs, io threads 16', T + ['-o', '/dev/shm/out.fq'], {'FXH_LANES': '2', 'FXH_IO_THREADS': '16'})
run('lanes=2 -> tmpfs, io threads 2', T + ['-o', '/dev/shm/out.fq'], {'FXH_LANES': '2', 'FXH_IO_THREADS': '2'})
run('lanes=2 -> tmpfs, 32 MB blocks', T + ['-o', '/dev/shm/out.fq'], {'FXH_LANES': '2', 'FXH_READ_BUFFER_MB': '32'})
run('lanes=2 -> tmpfs, 256 MB blocks', T + ['-o', '/dev/shm/out.fq'], {'FXH_LANES': '2', 'FXH_READ_BUFFER_MB': '256'})
run('lanes=1 -> tmpfs, 256 MB blocks', T + ['-o', '/dev/shm/out.fq'], {'FXH_LANES': '1', 'FXH_READ_BUFFER_MB': '256'})
run('host parse -> /dev/null', T + ['-o', '/dev/null'], {'FXH_HOST_PARSE': '1'})
t0 = time.perf_counter(); subprocess.run(['cp', '/dev/shm/in.fq', '/dev/shm/copy.fq']); print('cp 5.1 GB tmpfs->tmpfs: %.3f s' % (time.perf_counter() - t0))
t0 = time.perf_counter(); subprocess.run(['cat', '/dev/shm/in.fq'], stdout=subprocess.DEVNULL); print('cat 5.1 GB > /dev/null: %.3f s' % (time.perf_counter() - t0))
for f in ('in.fq', 'tiny.fq', 'out.fq', 'copy.fq'):
    try: os.unlink('/dev/shm/' + f)
    except OSError: pass
PY
echo "== gpu tests"
timeout 1500 python -m pytest tests -m gpu -q --durations=6 2>&1 | tail -16 | tee $O/pytest.txt
