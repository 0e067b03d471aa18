"""ctypes view of tests/emu/libfxgemu.so (serial CPU emulation of the tile kernels, test-only)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_EMU = os.path.join(_HERE, "emu")
_LIB = None
NCOUNTERS = 24


class Batch(C.Structure):
    _fields_ = [("bases", C.c_void_p), ("qual", C.c_void_p), ("len", C.c_void_p),
                ("fixed_len", C.c_uint32), ("stride", C.c_uint32), ("n", C.c_uint64)]


class Out(C.Structure):
    _fields_ = [("res", C.c_void_p), ("out_bases", C.c_void_p), ("out_qual", C.c_void_p), ("out_len", C.c_void_p),
                ("kept_index", C.c_void_p), ("out_off", C.c_void_p), ("counters", C.c_void_p)]


_CXX = ["hipcc", "--cuda-host-only", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wno-pass-failed", "-DFXG_HOST_EMULATION"]
_LINK = ["hipcc", "-shared", "-fPIC"]         # (objects only: with --cuda-host-only the driver would read them as sources)
_CSRC = os.path.join(_HERE, "..", "fastx_toolkit_amd", "csrc")
_EMU_DEPS = [os.path.join(_CSRC, f) for f in ("fxg_device.h", "fxg_kernels.h", "fxg_plan.h", "fxg_text.h", "fxg_rows.h", "fxg_history.h", "fxg_stats.h")]
EMU_UNITS = 8               # fxg_emu.cpp: -DFXG_EMU_TU=0 (everything but the clipper's instances) and its seven groups of clip instances


def _emu_objects(defs):
    """fxg_emu.cpp as EMU_UNITS objects compiled side by side (one unit took over three minutes); both libraries below link the same objects."""
    src = os.path.join(_EMU, "fxg_emu.cpp")
    d = os.path.join(_EMU, "obj%s" % "".join(x.replace("-D", "_").replace("=", "") for x in defs))
    os.makedirs(d, exist_ok=True)
    newest = max(os.path.getmtime(x) for x in [src] + _EMU_DEPS)
    objs = [os.path.join(d, "fxg_emu_%d.o" % k) for k in range(EMU_UNITS)]
    stale = [k for k, o in enumerate(objs) if not os.path.exists(o) or os.path.getmtime(o) < newest]
    procs = [(k, subprocess.Popen(_CXX + defs + ["-DFXG_EMU_TU=%d" % k, "-c", src, "-o", objs[k] + ".tmp"])) for k in stale]
    bad = [k for k, pr in procs if pr.wait() != 0]
    if bad:
        raise RuntimeError("emulator units %s did not compile" % bad)
    for k in stale:
        os.replace(objs[k] + ".tmp", objs[k])
    return objs


def build():
    defs = os.environ.get("FXG_EMU_DEFS", "").split()           # e.g. "-DFXG_V_TABLE": check a kernel variant on the CPU tier
    so = os.path.join(_EMU, "libfxgemu%s.so" % "".join(d.replace("-D", "_") for d in defs))
    objs = _emu_objects(defs)
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(o) for o in objs):
        subprocess.check_call(_LINK + objs + ["-o", so + ".tmp"])
        os.replace(so + ".tmp", so)
    return so


def build_stub():
    """tests/emu/stub/libfxg.so: the test-only look-alike of the engine library (fxg_stub.cpp over the emulator, plus the product's own
    multi-GPU host code, csrc/fxg_comm.h).  Returns the directory to put first on LD_LIBRARY_PATH."""
    build()
    d = os.path.join(_EMU, "stub")
    os.makedirs(d, exist_ok=True)
    so = os.path.join(d, "libfxg.so")
    src = os.path.join(_EMU, "fxg_stub.cpp")
    objs = _emu_objects([])
    deps = [src, os.path.join(_CSRC, "fxg_comm.h"), os.path.join(_HERE, "..", "include", "fxg.h")] + _EMU_DEPS + objs
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(x) for x in deps):
        subprocess.check_call(_CXX + ["-c", src, "-o", os.path.join(d, "fxg_stub.o")])
        subprocess.check_call(_LINK + [os.path.join(d, "fxg_stub.o")] + objs + ["-o", so + ".tmp", "-ldl"])
        os.replace(so + ".tmp", so)
    return d


def build_fake_rccl():
    """tests/emu/fakerccl/librccl.so.1: the five NCCL entry points the transport binds, over a shared-memory file (fake_rccl.c)."""
    d = os.path.join(_EMU, "fakerccl")
    os.makedirs(d, exist_ok=True)
    so, src = os.path.join(d, "librccl.so.1"), os.path.join(_EMU, "fake_rccl.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O1", "-std=gnu11", "-Wall", "-Wextra", "-fPIC", "-shared", src, "-o", so, "-ldl"])
    return d


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.fxg_emu_run_pipeline.argtypes = [C.POINTER(Batch), C.c_void_p, C.POINTER(Out), C.c_char_p, C.c_size_t]
        _LIB.fxg_emu_tile_reads.restype = C.c_uint
        _LIB.fxg_emu_run_pipeline_hist.argtypes = [C.POINTER(Batch), C.c_void_p, C.POINTER(Out), C.c_char_p, C.c_size_t, C.c_void_p]
        _LIB.fxg_emu_hist_new.restype = C.c_void_p
        _LIB.fxg_emu_hist_free.argtypes = [C.c_void_p]
        _LIB.fxg_emu_run_quality_stats.argtypes = [C.POINTER(Batch), C.c_void_p, C.c_uint32]
        _LIB.fxg_emu_quality_stats_piece_trips.restype = C.c_uint64
        _LIB.fxg_emu_quality_stats_piece_moved.restype = C.c_uint64
    return _LIB


def _aligned(n, dtype=np.uint8):
    """16-byte aligned zeroed array of n items."""
    item = np.dtype(dtype).itemsize
    raw = np.zeros(n * item + 32, dtype=np.uint8)
    off = (-raw.ctypes.data) % 16
    return raw[off:off + n * item].view(dtype)


def hist_new():
    return lib().fxg_emu_hist_new()


def hist_free(h):
    lib().fxg_emu_hist_free(h)


def run_pipeline(bases, qual, lens, params, fixed_len=None, compact=True, hist=None):
    n, stride = bases.shape
    b = _aligned(n * stride); b[:] = bases.reshape(-1)
    q = None
    if qual is not None:
        q = _aligned(n * stride); q[:] = qual.reshape(-1)
    if lens is not None:
        lens = np.ascontiguousarray(lens, dtype=np.uint16)
    res = np.zeros(n, dtype=np.uint32)
    ob, oq = _aligned(n * stride + 16), _aligned(n * stride + 16)
    ol, ki, oo = np.zeros(n, dtype=np.uint16), np.zeros(n, dtype=np.uint32), np.zeros(n, dtype=np.uint64)
    ctr = np.zeros(NCOUNTERS, dtype=np.uint64)
    bt = Batch(b.ctypes.data, q.ctypes.data if q is not None else None, lens.ctypes.data if lens is not None else None,
               int(fixed_len or stride), stride, n)
    o = Out(res.ctypes.data, ob.ctypes.data if compact else None, oq.ctypes.data if (compact and q is not None) else None,
            ol.ctypes.data, ki.ctypes.data, oo.ctypes.data, ctr.ctypes.data)
    err = C.create_string_buffer(512)
    rc = lib().fxg_emu_run_pipeline_hist(C.byref(bt), C.addressof(params), C.byref(o), err, 512, hist)
    if rc != 0:
        raise ValueError("emu rc=%d: %s" % (rc, err.value.decode()))
    kept, nbytes = int(ctr[1]), int(ctr[2])
    return dict(res=res, out_bases=ob[:nbytes].copy(), out_qual=oq[:nbytes].copy() if q is not None else None,
                out_len=ol[:kept], kept_index=ki[:kept], out_off=oo[:kept], counters=ctr)


def run_quality_stats(bases, qual, lens, fixed_len=None, hist=None, cols=None):
    """Adds the batch to hist[cols][5][128] (uint64) and returns it."""
    n, stride = bases.shape
    cols = cols or stride
    if hist is None:
        hist = np.zeros((cols, 5, 128), dtype=np.uint64)
    b = _aligned(n * stride); b[:] = bases.reshape(-1)
    q = None
    if qual is not None:
        q = _aligned(n * stride); q[:] = qual.reshape(-1)
    if lens is not None:
        lens = np.ascontiguousarray(lens, dtype=np.uint16)
    bt = Batch(b.ctypes.data, q.ctypes.data if q is not None else None, lens.ctypes.data if lens is not None else None,
               int(fixed_len or stride), stride, n)
    rc = lib().fxg_emu_run_quality_stats(C.byref(bt), hist.ctypes.data, hist.shape[0])
    if rc != 0:
        raise ValueError("emu quality_stats rc=%d" % rc)
    return hist


def last_plan_clip_global():
    """(clip_global, tile_reads) of the plan the last emulated run was made with (fxg_plan.h: the DP over the batch instead of a staged tile)"""
    a, t = C.c_int(), C.c_int()
    lib().fxg_emu_last_plan_clip_global(C.byref(a), C.byref(t))
    return bool(a.value), t.value


def last_plan():
    """(clip instance, runs its scratch-checkpoint two-pass form) of the last run_pipeline call -- what fxg_make_plan chose."""
    a, t = C.c_int(), C.c_int()
    lib().fxg_emu_last_plan(C.byref(a), C.byref(t))
    return a.value, bool(t.value)
