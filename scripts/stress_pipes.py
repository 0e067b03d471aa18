#!/usr/bin/env python3
"""CPU: chains of three tools through pipes (`trimmer | reverse_complement | filter`, over the emulation stub) with random block sizes, pipe capacities, numbers of
writer / reader threads through private pipes and lanes, against the same chain with one copying thread a side and untouched pipes.  `python scripts/stress_pipes.py <seed> <rounds>`."""
import os, subprocess, sys, hashlib, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import fxoracle_py as fo
import emu_py
stub = emu_py.build_stub()
tools = os.path.join(ROOT, 'fastx_toolkit_amd', 'host', 'bin')
import tempfile
TD = tempfile.mkdtemp()
INP = os.path.join(TD, 'in.fq')
rnd = random.Random(int(sys.argv[1]))
n_ok = 0
for it in range(int(sys.argv[2])):
    nreads = rnd.choice([30000, 90000, 200000])
    L = rnd.choice([36, 100, 151])
    text = fo.synth_fastq(rnd.randrange(1, 1 << 20), 0, nreads, L, False)
    open(INP, 'wb').write(text)
    base = dict(os.environ, LD_LIBRARY_PATH=stub, FXH_THREADS='2')
    t1 = [tools + '/fastq_quality_trimmer', '-t', '20', '-l', '30']
    t2 = [tools + '/fastx_reverse_complement']
    t3 = [tools + '/fastq_quality_filter', '-q', '20', '-p', '80']
    def chain(env):
        p1 = subprocess.Popen(t1 + ['-i', INP], stdout=subprocess.PIPE, env=env)
        p2 = subprocess.Popen(t2, stdin=p1.stdout, stdout=subprocess.PIPE, env=env)
        p3 = subprocess.Popen(t3, stdin=p2.stdout, stdout=subprocess.PIPE, env=env)
        p1.stdout.close(); p2.stdout.close()
        o = p3.communicate()[0]
        assert p1.wait() == 0 and p2.wait() == 0 and p3.returncode == 0
        return hashlib.md5(o).hexdigest(), len(o)
    want = chain(dict(base, FXH_NO_PIPE_FANOUT='1', FXH_NO_PIPE_TUNING='1'))
    for k in range(3):
        e = dict(base, FXH_READ_BUFFER_MB=str(rnd.choice([1, 2, 4, 16])), FXH_PIPE_MB=str(rnd.choice([1, 1, 2, 8])), FXH_PIPE_READERS=str(rnd.randrange(2, 9)), FXH_PIPE_WRITERS=str(rnd.randrange(2, 5)),
                 FXH_LANES=str(rnd.randrange(1, 4)))
        if rnd.random() < 0.3: e['FXH_NO_PIPE_TUNING'] = '1'
        got = chain(e)
        assert got == want, (it, e, got, want)
        n_ok += 1
print('seed', sys.argv[1], 'chains', n_ok, 'ok')
