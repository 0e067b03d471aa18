"""Build the native pieces in-tree (no JIT cache): libfxg.so (HIP, gfx950) and the C host tools."""
import os
import shutil
import subprocess
import sys

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


def check_exec_zero(so):
    """Refuse a library whose kernels contain the miscompile of DESIGN.md section 3: ROCm 7.2's register allocator can put the spill
    stores / copies of a lane-divergent loop's live-out values into the loop's exit block AHEAD of the EXEC restore, where they run
    for no lane -- silently wrong integers that move with every change of the allocation (scripts/check_exec_zero.py reads the ISA)."""
    tool = os.path.join(ROOT, "scripts", "check_exec_zero.py")
    p = subprocess.run([sys.executable, tool, so], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if p.returncode not in (0, 1):          # 2 (or a crash): the check itself failed -- no tools, no code object, a listing it cannot read
        raise RuntimeError("%s NOT CHECKED: scripts/check_exec_zero.py could not read the library's ISA (exit code %d); that is a broken check, "
                           "not a finding, and a library that was not checked is not accepted.\n%s" % (so, p.returncode, p.stdout[-3000:]))
    if p.returncode != 0:
        raise RuntimeError("%s REJECTED: vector/memory code runs with EXEC = 0 at the exit of a lane-divergent loop (compiler bug, DESIGN.md section 3).\n"
                           "Change the register budget of the named instance (fxg_clip_waves / __launch_bounds__) or its source and rebuild.\n%s" % (so, p.stdout[-3000:]))


CLIP_UNITS = 7      # csrc/fxg_engine_clip.hip is compiled once per group of clip instances (-DFXG_CLIP_TU=1..7), beside csrc/fxg_engine.hip (-DFXG_SPLIT)


def compile_engine(out, extra_flags=(), check=True):
    """The engine into `out`, through a temporary name: a library that fails the ISA check never appears under `out`.  Eight translation units compiled
    side by side (csrc/fxg_host.h: the clip instances are most of the work; as one unit the build took three and a half minutes), then one link."""
    tmp = out + ".new"
    objdir = os.path.join(ROOT, "build", "engine_" + os.path.basename(out).replace(".", "_"))
    os.makedirs(objdir, exist_ok=True)
    cflags = [f for f in HIPCC_FLAGS if f != "-shared"] + list(extra_flags) + ["-c"]
    units = [("fxg_engine.hip", ["-DFXG_SPLIT"], os.path.join(objdir, "engine.o"))]
    units += [("fxg_engine_clip.hip", ["-DFXG_CLIP_TU=%d" % k], os.path.join(objdir, "clip%d.o" % k)) for k in range(1, CLIP_UNITS + 1)]
    procs = [(src, subprocess.Popen([hipcc()] + cflags + defs + [os.path.join(CSRC, src), "-o", obj])) for src, defs, obj in units]
    failed = [src for src, p in procs if p.wait() != 0]
    if failed:
        raise RuntimeError("hipcc failed on %s" % ", ".join(failed))
    subprocess.check_call([hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared"] + [obj for _, _, obj in units] + ["-o", tmp])
    if check:
        try:
            check_exec_zero(tmp)
        except Exception:
            os.replace(tmp, out + ".rejected")
            raise
    os.replace(tmp, out)
    return out


def build_engine(force=False):
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(ROOT, "include", "fxg.h")]
    if force or _newer(LIBFXG, deps):
        compile_engine(LIBFXG)
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
