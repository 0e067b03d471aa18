"""CPU tier: the product's RCCL transport (csrc/fxg_comm.h: fxg_comm_create / fxg_epilogue_rccl) with world = 2, 3 and 8.

The transport is host code; the emulation stub compiles the very same header over its host-memory "device", and tests/emu/fake_rccl.c
stands in for librccl.so.1 (the five NCCL entry points over a shared-memory file).  One process per rank, as on a node: rendezvous
through the id file (written once, by rank 0; removed once the communicator is up), a late rank 0, a rank that never finds the id,
rank order of the gathered blocks, offsets = exclusive scan of the kept counts, the error paths of every NCCL call.
The real library runs the same code against the real RCCL in tests/test_gpu_parity.py::test_rccl_epilogue_*.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

NCOUNTERS = 24
WORKER = r"""
import ctypes as C, json, os, sys, time
so, idfile, rank, world, timeout, delay = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), float(sys.argv[6])
L = C.CDLL(so)
L.fxg_last_error.restype = C.c_char_p
L.fxg_last_error.argtypes = [C.c_void_p]
L.fxg_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
L.fxg_comm_create.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_void_p)]
L.fxg_epilogue_rccl.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_void_p]
L.fxg_comm_destroy.argtypes = [C.c_void_p]
ctx = C.c_void_p()
assert L.fxg_ctx_create(0, C.byref(ctx)) == 0
time.sleep(delay)
comm = C.c_void_p()
rc = L.fxg_comm_create(ctx, idfile.encode(), rank, world, timeout, C.byref(comm))
out = dict(rank=rank, create_rc=rc, error=L.fxg_last_error(ctx).decode())
if rc == 0:
    out["idfile_after_create"] = os.path.exists(idfile)
    rounds = []
    for rnd in range(2):                                        # two passes over one communicator
        mine = (C.c_uint64 * 24)(*[1000 * (rank + 1) + 10 * rnd + i for i in range(24)])    # this rank's counter block ("device" memory of the stub)
        mine[15] = 1 << rank                                    # FXG_C_ERRORS: OR-ed, not added
        totals, gathered = (C.c_uint64 * 24)(), (C.c_uint64 * (24 * world))()
        ro, bo = C.c_uint64(), C.c_uint64()
        erc = L.fxg_epilogue_rccl(ctx, comm, mine, totals, C.byref(ro), C.byref(bo), gathered)
        rounds.append(dict(rc=erc, error=L.fxg_last_error(ctx).decode() if erc else "", totals=list(totals), gathered=list(gathered), read_off=ro.value, byte_off=bo.value))
    out["rounds"] = rounds
    L.fxg_comm_destroy(comm)
print(json.dumps(out))
"""


@pytest.fixture(scope="module")
def libs():
    import emu_py
    return os.path.join(emu_py.build_stub(), "libfxg.so"), emu_py.build_fake_rccl()


def _spawn(libs, tmp_path, world, ranks=None, timeout=20, delays=None, env_extra=None, tag="job"):
    so, fake = libs
    idfile = str(tmp_path / ("%s.id" % tag))
    log = str(tmp_path / ("%s.log" % tag))
    env = dict(os.environ, LD_LIBRARY_PATH=fake + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""), FXG_FAKE_RCCL_LOG=log)
    env.update(env_extra or {})
    procs = [subprocess.Popen([sys.executable, "-c", WORKER, so, idfile, str(r), str(world), str(timeout), str((delays or {}).get(r, 0.0))],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env) for r in (ranks if ranks is not None else range(world))]
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=120)
        assert p.returncode == 0, e.decode()[-2000:]
        outs.append(json.loads(o.decode().splitlines()[-1]))
    return sorted(outs, key=lambda d: d["rank"]), idfile, (open(log).read() if os.path.exists(log) else "")


def _block(rank, rnd):
    b = [1000 * (rank + 1) + 10 * rnd + i for i in range(NCOUNTERS)]
    b[15] = 1 << rank
    return b


@pytest.mark.parametrize("world", [2, 3, 8])
def test_transport_world_n(libs, tmp_path, world):
    outs, idfile, log = _spawn(libs, tmp_path, world, delays={0: 0.6})      # rank 0 is LATE: the others poll for the id file meanwhile
    assert log.count("getid") == 1, log                                          # the id is made once, by rank 0
    assert sorted(l for l in log.splitlines() if l.startswith("init")) == sorted("init rank %d world %d" % (r, world) for r in range(world))
    assert not os.path.exists(idfile) and not any(f.startswith(os.path.basename(idfile)) for f in os.listdir(tmp_path) if f.endswith(".tmp"))
    for r, o in enumerate(outs):
        assert o["create_rc"] == 0, o
        for rnd, rd in enumerate(o["rounds"]):
            assert rd["rc"] == 0, rd
            blocks = [_block(g, rnd) for g in range(world)]
            assert rd["gathered"] == [x for b in blocks for x in b]              # rank order
            tot = np.array(blocks, dtype=np.uint64).sum(axis=0)
            tot[15] = (1 << world) - 1
            assert rd["totals"] == [int(x) for x in tot]
            assert rd["read_off"] == sum(b[1] for b in blocks[:r]) and rd["byte_off"] == sum(b[2] for b in blocks[:r])      # exclusive scan of kept reads / kept bases
    # rank 0 removed the id file once the communicator was up: a rank that turns up for the same name later cannot pick up a stale id
    assert outs[0]["idfile_after_create"] is False


def test_rank_without_rank0_times_out(libs, tmp_path):
    outs, idfile, log = _spawn(libs, tmp_path, 2, ranks=[1], timeout=1)
    assert outs[0]["create_rc"] != 0 and "no RCCL id in" in outs[0]["error"] and "after 1 s" in outs[0]["error"], outs
    assert "init" not in log


def test_truncated_id_file_is_not_an_id(libs, tmp_path):
    """A reader sees a whole record or nothing: a short file (a writer that is not the transport's rename) is never taken for an id."""
    (tmp_path / "job.id").write_bytes(b"x" * 17)
    outs, _, log = _spawn(libs, tmp_path, 2, ranks=[1], timeout=1)
    assert outs[0]["create_rc"] != 0 and "no RCCL id" in outs[0]["error"] and "init" not in log


def _record(world, job_token, fill=b"S"):
    h = 0
    if job_token:
        h = 0xcbf29ce484222325
        for ch in job_token.encode():
            h = ((h ^ ch) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    import struct
    return b"FXGRDV1\0" + struct.pack("<IIQ", world, 0, h) + fill * 128


@pytest.mark.parametrize("stale,env", [((3, None), {}), ((2, "job-of-yesterday"), {"FXG_COMM_JOB": "job-of-today"}), ((2, None), {"FXG_COMM_JOB": "job-of-today"})])
def test_record_of_a_dead_job_is_not_taken(libs, tmp_path, stale, env):
    """A job that died between publishing its id and removing it leaves a record under the name.  The next job's early ranks must not join on it:
    a record for another world size, or with another job token (FXG_COMM_JOB), is not an id for this job -- the ranks wait for their own rank 0,
    which removes the leftover before it makes its id (here it is 0.6 s late on purpose)."""
    (tmp_path / "job.id").write_bytes(_record(*stale))
    outs, idfile, log = _spawn(libs, tmp_path, 2, delays={0: 0.6}, env_extra=env)
    assert log.count("getid") == 1 and sorted(l for l in log.splitlines() if l.startswith("init")) == ["init rank 0 world 2", "init rank 1 world 2"], log
    assert all(o["create_rc"] == 0 and o["rounds"][0]["rc"] == 0 for o in outs), outs
    assert outs[0]["rounds"][0]["gathered"] == outs[1]["rounds"][0]["gathered"] == _block(0, 0) + _block(1, 0)
    assert not os.path.exists(idfile)
    # and with nobody to replace it, the leftover alone never becomes an id
    (tmp_path / "job2.id").write_bytes(_record(*stale))
    outs, _, log = _spawn(libs, tmp_path, 2, ranks=[1], timeout=1, env_extra=env, tag="job2")
    assert outs[0]["create_rc"] != 0 and "no RCCL id" in outs[0]["error"] and "init" not in log


@pytest.mark.parametrize("what,needle", [("id", "ncclGetUniqueId: internal error (fake)"), ("init", "ncclCommInitRank(rank 0 of 1): internal error (fake)"),
                                         ("gather", "ncclAllGather: internal error (fake)")])
def test_nccl_errors_are_reported(libs, tmp_path, what, needle):
    outs, idfile, _ = _spawn(libs, tmp_path, 1, env_extra={"FXG_FAKE_RCCL_FAIL": what})
    o = outs[0]
    if what == "gather":
        assert o["create_rc"] == 0 and o["rounds"][0]["rc"] != 0 and needle in o["rounds"][0]["error"], o
    else:
        assert o["create_rc"] != 0 and needle in o["error"], o
    assert not os.path.exists(idfile)


def test_gather_buffer_allocation_failure(libs, tmp_path):
    outs, idfile, _ = _spawn(libs, tmp_path, 1, env_extra={"FXG_EMU_COMM_NOMEM": "1"})
    assert outs[0]["create_rc"] != 0 and "gather buffer" in outs[0]["error"] and not os.path.exists(idfile)


def test_unwritable_rendezvous_path(libs, tmp_path):
    so, fake = libs
    env = dict(os.environ, LD_LIBRARY_PATH=fake)
    p = subprocess.run([sys.executable, "-c", WORKER, so, str(tmp_path / "no" / "such" / "dir" / "id"), "0", "1", "1", "0"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=60)
    o = json.loads(p.stdout.decode().splitlines()[-1])
    assert o["create_rc"] != 0 and "cannot write the rendezvous file" in o["error"]
