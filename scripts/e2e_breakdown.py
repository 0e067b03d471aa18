#!/usr/bin/env python3
"""Where the end-to-end wall time of a tool goes (GPU box): FXH_TIMING phases under different lane / writer / exit settings."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from concurrent.futures import ThreadPoolExecutor
from oracle import fxoracle_py as fo
R = int(sys.argv[1]) if len(sys.argv) > 1 else 16_000_000
chunk = 250_000
with ThreadPoolExecutor(max_workers=32) as ex:
    parts = list(ex.map(lambda k: fo.synth_fastq(2, k * chunk, chunk, 150, False), range(R // chunk)))
with open('/dev/shm/in.fq', 'wb') as f:
    for p in parts: f.write(p)
with open('/dev/shm/tiny.fq', 'wb') as f: f.write(b'\n'.join(parts[0].split(b'\n')[:4000]) + b'\n')
del parts
B = os.path.join(ROOT, 'fastx_toolkit_amd/host/bin/')
ONLY = os.environ.get('ONLY')            # run only the lines whose label contains this
def run(label, argv, env=None):
    if ONLY and ONLY not in label: return
    e = dict(os.environ, FXH_TIMING='1', **(env or {}))
    best = None
    for _ in range(3):
        for o in ('/dev/shm/out.fq',):                     # freeing the pages of the previous output is not the tool's time
            try: os.unlink(o)
            except OSError: pass
        t0 = time.perf_counter(); p = subprocess.run(argv, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e); dt = time.perf_counter() - t0
        if p.returncode != 0:
            print(label, 'FAILED', p.stderr[-300:]); return
        if best is None or dt < best[0]: best = (dt, p.stderr.decode(errors='replace') + p.stdout.decode(errors='replace'))
    tl = ' | '.join(l for l in best[1].splitlines() if l.startswith('fxh timing'))
    print('%-46s wall %.3f s = %5.1f Mreads/s   %s' % (label, best[0], R / best[0] / 1e6, tl[tl.find('run'):] if 'run' in tl else tl), flush=True)
T = [B + 'fastq_quality_trimmer', '-t', '20', '-l', '30', '-i', '/dev/shm/in.fq']
TF = [B + 'fastq_quality_trim_filter', '-t', '20', '-l', '30', '-q', '20', '-p', '80', '-i', '/dev/shm/in.fq']
run('tiny input (1000 reads): startup', [B + 'fastq_quality_trimmer', '-t', '20', '-l', '30', '-i', '/dev/shm/tiny.fq', '-o', '/dev/shm/out.fq'])
run('tiny input, slow exit', [B + 'fastq_quality_trimmer', '-t', '20', '-l', '30', '-i', '/dev/shm/tiny.fq', '-o', '/dev/shm/out.fq'], {'FXH_SLOW_EXIT': '1', 'FXH_TEARDOWN': '1'})
for lanes in ('1', '2', '3'):
    run('trimmer lanes=%s -> tmpfs file' % lanes, T + ['-o', '/dev/shm/out.fq'], {'FXH_LANES': lanes})
run('trimmer lanes=2 -> /dev/null', T + ['-o', '/dev/null'], {'FXH_LANES': '2'})
run('trimmer lanes=2 -> tmpfs, slow exit + teardown', T + ['-o', '/dev/shm/out.fq'], {'FXH_LANES': '2', 'FXH_SLOW_EXIT': '1', 'FXH_TEARDOWN': '1'})
for io in ('2', '4', '16'):
    run('trimmer lanes=2 -> tmpfs, io threads %s' % io, T + ['-o', '/dev/shm/out.fq'], {'FXH_LANES': '2', 'FXH_IO_THREADS': io})
run('trimmer lanes=2 -> tmpfs, 128 MB blocks', T + ['-o', '/dev/shm/out.fq'], {'FXH_LANES': '2', 'FXH_READ_BUFFER_MB': '128'})
run('trimmer lanes=2 -> tmpfs, 32 MB blocks', T + ['-o', '/dev/shm/out.fq'], {'FXH_LANES': '2', 'FXH_READ_BUFFER_MB': '32'})
run('fused trim+filter lanes=2 -> tmpfs', TF + ['-o', '/dev/shm/out.fq'], {'FXH_LANES': '2'})
run('fused trim+filter lanes=1 -> tmpfs', TF + ['-o', '/dev/shm/out.fq'], {'FXH_LANES': '1'})
run('fused, two contexts as two devices', TF + ['-o', '/dev/shm/out.fq'], {'FXG_DEVICES': '0,0', 'FXH_LANES': '1'})
run('host parse -> tmpfs', T + ['-o', '/dev/shm/out.fq'], {'FXH_HOST_PARSE': '1'})
for f in ('in.fq', 'tiny.fq', 'out.fq'):
    try: os.unlink('/dev/shm/' + f)
    except OSError: pass
