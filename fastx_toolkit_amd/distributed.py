"""One process per GPU; reads shard by contiguous index range, so concatenating the ranks' packed outputs
in rank order reproduces the single-process output order.  There is no exchange during
compute; the only collective is one all-gather of each rank's 16-slot counter block (128 bytes), from
which every rank derives the job totals and its own offset into the global kept-read / kept-byte stream.
Backend "nccl" is RCCL on ROCm (xGMI); "gloo" is used by the CPU tests.
"""
import os

import numpy as np

import ctypes as C

from .engine import NCOUNTERS, load_library


def shard_range(n_total, rank, world):
    """Reads [lo, hi) owned by `rank`: g*N/G .. (g+1)*N/G (fxg_shard_range of the C-ABI)."""
    lo, hi = C.c_uint64(), C.c_uint64()
    if load_library().fxg_shard_range(n_total, rank, world, C.byref(lo), C.byref(hi)) != 0:
        raise ValueError("shard_range(%r, %r, %r)" % (n_total, rank, world))
    return lo.value, hi.value


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun's environment; returns (rank, local_rank, world)."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = backend or os.environ.get("FXG_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def gather_counters(counters, group=None):
    """One all-gather of every rank's counter block (NCOUNTERS x int64 = 192 bytes).  Asynchronous for RCCL:
    the result stays on the device, nothing is copied to the host.  Returns a [world, NCOUNTERS] tensor."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return counters.reshape(1, -1)
    world = dist.get_world_size(group)
    if dist.get_backend(group) == "gloo":      # CPU tests / shared-GPU smoke runs: stage through host memory
        counters = counters.detach().cpu()
    if counters.is_cuda:
        gathered = torch.empty((world, NCOUNTERS), dtype=counters.dtype, device=counters.device)
        dist.all_gather_into_tensor(gathered, counters.contiguous(), group=group)
        return gathered
    parts = [torch.empty(NCOUNTERS, dtype=counters.dtype) for _ in range(world)]
    dist.all_gather(parts, counters.contiguous(), group=group)
    return torch.stack(parts)


def offsets_from_gathered(gathered, rank):
    """(totals uint64[NCOUNTERS], kept_read_offset, kept_byte_offset, per_rank) from gather_counters' result (host side)."""
    per_rank = np.ascontiguousarray(gathered.detach().cpu().numpy().view(np.uint64).reshape(-1, NCOUNTERS))
    totals = np.zeros(NCOUNTERS, dtype=np.uint64)
    read_off, byte_off = C.c_uint64(), C.c_uint64()
    rc = load_library().fxg_epilogue(per_rank.ctypes.data, per_rank.shape[0], rank, totals.ctypes.data, C.byref(read_off), C.byref(byte_off))
    if rc != 0:
        raise ValueError("fxg_epilogue failed (%d ranks, rank %d)" % (per_rank.shape[0], rank))
    return totals, read_off.value, byte_off.value, per_rank


def concat_pwrite(fd, arr, offset):
    """Write this rank's packed slice (a contiguous numpy array) at its offset of the job's output file (fxg_concat_pwrite)."""
    arr = np.ascontiguousarray(arr)
    if load_library().fxg_concat_pwrite(fd, arr.ctypes.data, arr.nbytes, offset) != 0:
        raise OSError("fxg_concat_pwrite failed")


def epilogue(counters, group=None):
    """counters: int64[NCOUNTERS] tensor of this rank (device tensor for RCCL, CPU tensor for gloo).

    Returns (totals uint64[NCOUNTERS], kept_read_offset, kept_byte_offset, per_rank uint64[world, NCOUNTERS]).
    """
    import torch.distributed as dist
    rank = dist.get_rank(group) if (dist.is_available() and dist.is_initialized()) else 0
    return offsets_from_gathered(gather_counters(counters, group), rank)
