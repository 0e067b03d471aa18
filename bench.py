#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X FASTQ engine (driver contract in the task statement).

Workload (BASELINE.json configs[1], the configuration the metric is quoted on):
    fastq_quality_trimmer -t 20 -l 30 | fastq_quality_filter -q 20 -p 80
on 50 M synthetic 150 bp Phred+33 reads PER GPU (seed 2, SURVEY.md 8d generator, generated on the
device so inputs are resident in HBM when the timed region starts), fused into ONE pass that also
stream-compacts the kept, trimmed reads in input order.  One "step" = one such pass over the batch.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--reads R]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

Multi-GPU: reads shard by contiguous index range (rank g owns reads [g*R, (g+1)*R) of the global set),
no data-path collective; each step ends with one 192-byte all-gather of the counter blocks (RCCL) from
which every rank derives the job totals and its offset in the global output.  scaling = weak.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy ceiling
READ_LEN = 150
SEED = 2
PARAMS = dict(stages=2 | 4, qt_threshold=20, qt_min_len=30, qf_min_quality=20, qf_min_percent=80)


def cpu_baseline(reads_per_pipe=250_000, max_pipes=16):
    """Reference CPU path on this box's host cores, bounded sample of the same workload (rank 0, N=1 only).

    SURVEY 8d: the reference is single-threaded, so the input is split at record boundaries into P chunks and P
    `trimmer | filter` shell pipes (2 processes each) run concurrently; the aggregate rate over 2P cores is reported.
    """
    from oracle import fxoracle_py as fo
    ref = fo.ref_binary()
    try:
        fo.lib()
    except Exception:
        return None
    ncpu = os.cpu_count() or 2
    pipes = max(1, min(max_pipes, ncpu // 2))
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as td:
        if ref:
            files = []
            for k in range(pipes):
                inp = os.path.join(td, "in%d.fq" % k)
                with open(inp, "wb") as f:
                    f.write(fo.synth_fastq(SEED, k * reads_per_pipe, reads_per_pipe, READ_LEN, False))
                files.append(inp)
            t0 = time.perf_counter()
            procs = []
            for k, inp in enumerate(files):
                p1 = subprocess.Popen([ref, "fastq_quality_trimmer", "-t", "20", "-l", "30", "-i", inp], stdout=subprocess.PIPE)
                p2 = subprocess.Popen([ref, "fastq_quality_filter", "-q", "20", "-p", "80", "-o", os.path.join(td, "out%d.fq" % k)], stdin=p1.stdout)
                p1.stdout.close()
                procs += [p1, p2]
            ok = all(p.wait() == 0 for p in procs)
            dt = time.perf_counter() - t0
            if ok:
                n = pipes * reads_per_pipe
                return dict(value=round(n / dt / 1e6, 4), unit="Mreads/s", cores=2 * pipes, kind="reference",
                            sample="first %d reads of the same seed-2 150 bp set as FASTQ text on tmpfs, split into %d chunks; each chunk piped "
                                   "through the reference libfastx reader/writer (compiled -O3 from /root/reference/src/libfastx) with the "
                                   "trimmer|filter loop bodies of oracle/ref_driver.cpp; %d concurrent single-threaded processes = %d cores"
                                   % (n, pipes, 2 * pipes, 2 * pipes))
        # fall back to the plain-C port (SoA in memory, no text I/O), 1 thread
        n = 4_000_000
        b, q = fo.synth_batch(SEED, 0, n, READ_LEN)
        t0 = time.perf_counter()
        fo.run_pipeline(b, q, None, fo.make_params(**PARAMS))
        dt = time.perf_counter() - t0
        return dict(value=round(n / dt / 1e6, 4), unit="Mreads/s", cores=1, kind="port",
                    sample="oracle/fxoracle.c on %d in-memory SoA reads (no FASTQ text parsing/formatting), 1 thread" % n)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--reads", type=int, default=50_000_000, help="reads per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--decision-only", action="store_true", help="no compaction: 154 B/read variant (not the headline)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from fastx_toolkit_amd import Engine, make_params
    from fastx_toolkit_amd import distributed as fxd

    rank, local, world = fxd.init_from_env()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run" % (args.gpus, world))
    if os.environ.get("FXG_BENCH_SHARED_GPU"):      # smoke-testing the N>1 path on a 1-GPU box (with FXG_DIST_BACKEND=gloo)
        local = 0
    torch.cuda.set_device(local)
    # One stream for everything: torch allocations/fills, the engine's kernels (it adopts torch's current stream) and the
    # RCCL all-gather, which orders itself after that stream -- so the gather reads the counters of the pass just enqueued.
    torch.cuda.set_stream(torch.cuda.Stream(device=local))
    eng = Engine(local)
    R, L = args.reads, READ_LEN
    lo = rank * R                                  # weak scaling: rank g owns reads [g*R, (g+1)*R) of the global set
    bases, qual = eng.synth(SEED, lo, R, L, False)
    params = make_params(**PARAMS)
    compact = not args.decision_only
    outs = eng.alloc_outputs(R, L, compact=compact, meta=False)
    torch.cuda.synchronize()

    gathered = [None]

    def step():
        r = eng.run(bases, qual, params, fixed_len=L, compact=compact, meta=False, outputs=outs)
        if world > 1:
            gathered[0] = fxd.gather_counters(outs["counters"])   # 192-byte all-gather, enqueued after the kernels, no host sync
        return r

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=eng.device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    counters = res.counters
    kept, kept_bytes = int(counters[1]), int(counters[2])
    if world > 1:                                  # job totals and this rank's offsets in the global output, from the last step's gather
        totals, read_off, byte_off, _ = fxd.offsets_from_gathered(gathered[0], rank)
        assert int(totals[0]) == R * world

    # kernel-level time of the dominant kernel: HIP events on the launch stream, one launch at a time
    eng.set_profiling(True)
    kms = []
    for _ in range(max(5, min(args.steps, 20))):
        eng.run(bases, qual, params, fixed_len=L, compact=compact, meta=False, outputs=outs)
        kms.append(eng.last_kernel_ms())
    eng.set_profiling(False)
    kavg = sum(kms) / len(kms)
    launch = eng.last_launch()
    # algorithmic bytes per launch: read 2L per read, write 4 B result per read + 2*new_len per kept read
    alg_bytes = R * (2 * L + 4) + 2 * kept_bytes if compact else R * (L + 4)
    achieved = alg_bytes / (kavg * 1e-3) / 1e9
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc):
        try:
            pj = json.load(open(pmc))
            traffic = pj.get("hbm_bytes_per_launch") if (pj.get("reads_per_launch") == R and compact) else None
        except Exception:
            traffic = None

    if rank == 0:
        total_reads = R * world * args.steps
        out = {
            "metric": "Mreads/s (150 bp) quality-trim+filter",
            "value": round(total_reads / dt / 1e6, 2),
            "unit": "Mreads/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {
                "workload": "cfg2: fastq_quality_trimmer -t 20 -l 30 | fastq_quality_filter -q 20 -p 80, %d x %d bp Phred+33 reads per GPU, "
                            "%s" % (R, L, "one fused pass with order-preserving compaction of the kept trimmed reads" if compact
                                    else "decision-only pass (no compaction)"),
                "reads_per_gpu": R, "read_len": L, "seed": SEED, "kept_reads_per_gpu": kept, "kept_bases_per_gpu": kept_bytes,
                "gbases_per_s_in": round(total_reads * L / dt / 1e9, 2), "parallelism": "reads sharded x%d, no data-path collective" % world,
            },
            "roofline": {
                "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "kernel": launch["kernel"], "kernel_ms_avg": round(kavg, 4), "kernel_ms_min": round(min(kms), 4),
                "algorithmic_bytes_per_launch": alg_bytes, "bytes_per_read": round(alg_bytes / R, 2),
                "grid": launch["grid"], "block": launch["block"], "lds_bytes": launch["lds"], "tile_reads": launch["tile_reads"],
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
