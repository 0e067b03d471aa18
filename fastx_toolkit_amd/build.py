"""Build the native pieces in-tree (no JIT cache): libfxg.so (HIP, gfx950) and the C host tools."""
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
HOST = os.path.join(PKG, "host")
LIBFXG = os.path.join(PKG, "libfxg.so")

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the gfx950 engine cannot be built")
    return exe


def build_engine(force=False):
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(ROOT, "include", "fxg.h")]
    if force or _newer(LIBFXG, deps):
        subprocess.check_call([hipcc()] + HIPCC_FLAGS + [os.path.join(CSRC, "fxg_engine.hip"), "-o", LIBFXG])
    return LIBFXG


def build_host(force=False):
    """C host layer: libfastx-compatible record API + the command-line tools (links libfxg.so and zlib)."""
    mk = os.path.join(HOST, "Makefile")
    if os.path.exists(mk):
        subprocess.check_call(["make", "-s", "-C", HOST] + (["-B"] if force else []))


def build_all(force=False):
    build_engine(force)
    build_host(force)
    return LIBFXG
