"""CPU tier: the N>1 path (contiguous shards + one all-gather of the counter blocks) under gloo, world_size 2.

Each rank runs the checker on its shard (standing in for the GPU engine, which needs a device), calls the
product's epilogue (fxg_shard_range / fxg_epilogue of the C-ABI under torch.distributed's all-gather), and writes its
packed slice with the product's fxg_concat_pwrite at the offset the epilogue returned; the assembled file must equal the
single-process result.
"""
import os
import socket
import tempfile

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import fxoracle_py as fo

N, L, WORLD = 6001, 60, 2
PD = dict(stages=6, qt_threshold=20, qt_min_len=20, qf_min_quality=18, qf_min_percent=70)


def _worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from fastx_toolkit_amd import distributed as fxd
    r, _, w = fxd.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    lo, hi = fxd.shard_range(N, rank, world)
    b, q = fo.synth_batch(4, lo, hi - lo, L)
    o = fo.run_pipeline(b, q, None, fo.make_params(**PD))
    counters = torch.from_numpy(o["counters"].view(np.int64).copy())
    totals, read_off, byte_off, per_rank = fxd.epilogue(counters)
    # the concatenation is product code too: every rank writes its slice at the offset the epilogue returned
    fd = os.open(os.path.join(tmp, "bases.bin"), os.O_WRONLY)
    fxd.concat_pwrite(fd, o["out_bases"], byte_off)
    os.close(fd)
    fd = os.open(os.path.join(tmp, "kept.bin"), os.O_WRONLY)
    fxd.concat_pwrite(fd, o["kept_index"] + np.uint32(lo), read_off * 4)
    os.close(fd)
    np.save(os.path.join(tmp, "totals%d.npy" % rank), totals)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_epilogue():
    from fastx_toolkit_amd import distributed as fxd
    assert [fxd.shard_range(10, g, 4) for g in range(4)] == [(0, 2), (2, 5), (5, 7), (7, 10)]
    b, q = fo.synth_batch(4, 0, N, L)
    ref = fo.run_pipeline(b, q, None, fo.make_params(**PD))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    with tempfile.TemporaryDirectory() as tmp:
        for name, size in (("bases.bin", len(ref["out_bases"])), ("kept.bin", 4 * len(ref["kept_index"]))):
            with open(os.path.join(tmp, name), "wb") as f:
                f.write(b"\0" * size)
        mp.spawn(_worker, args=(WORLD, port, tmp), nprocs=WORLD, join=True)
        assert open(os.path.join(tmp, "bases.bin"), "rb").read() == ref["out_bases"].tobytes()
        assert np.array_equal(np.fromfile(os.path.join(tmp, "kept.bin"), dtype=np.uint32), ref["kept_index"])
        for g in range(WORLD):
            assert np.array_equal(np.load(os.path.join(tmp, "totals%d.npy" % g))[:13], ref["counters"][:13])


def test_single_process_epilogue_is_identity():
    from fastx_toolkit_amd import distributed as fxd
    c = torch.arange(24, dtype=torch.int64)
    totals, ro, bo, per = fxd.epilogue(c)
    assert ro == 0 and bo == 0 and list(totals) == list(range(24)) and per.shape == (1, 24)
