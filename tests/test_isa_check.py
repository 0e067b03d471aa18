"""CPU tier: the ISA gate of the build (scripts/check_exec_zero.py, fastx_toolkit_amd/build.py: check_exec_zero).

ROCm 7.2's register allocator can put the spill stores and copies of a lane-divergent loop's live-out values into the loop's exit
block ahead of the EXEC restore, where they run for no lane (how one clip instance came out wrong on the GPU
only).  The shipped library must be free of the pattern, the checker must see it in a listing that has it, and the build must refuse
such a library."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

TOOL = os.path.join(ROOT, "scripts", "check_exec_zero.py")
ISA = os.path.join(ROOT, "tests", "golden", "isa")


def _run(path):
    p = subprocess.run([sys.executable, TOOL, path], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    return p.returncode, p.stdout


def test_checker_sees_the_pattern_in_a_listing():
    rc, out = _run(os.path.join(ISA, "exec_zero_bad.dis"))
    assert rc == 1 and "scratch_store_dwordx2 off, v[192:193], off offset:280" in out and "v_mov_b32_e32 v12, v248" in out
    assert "v_readlane" not in out                              # ignores EXEC: not part of the report
    rc, out = _run(os.path.join(ISA, "exec_zero_good.dis"))     # the same instructions behind the restore
    assert rc == 0, out


def test_shipped_library_is_clean_and_a_bad_one_is_refused(tmp_path):
    from fastx_toolkit_amd import build as b
    b.build_engine()
    rc, out = _run(b.LIBFXG)
    assert rc == 0, out
    assert "kernels, 0 places" in out
    b.check_exec_zero(b.LIBFXG)
    with pytest.raises(RuntimeError, match="REJECTED"):
        b.check_exec_zero(os.path.join(ISA, "exec_zero_bad.dis"))


def test_check_fails_closed(tmp_path):
    """A check that could not read the ISA is not a clean library: no code object, a listing the parser finds no kernels in, a missing file --
    exit code 2 from the tool and NOT CHECKED (not REJECTED, not accepted) from the build."""
    from fastx_toolkit_amd import build as b
    empty = tmp_path / "nothing.dis"
    empty.write_text("this is not a disassembly\n")
    odd = tmp_path / "odd_format.dis"                         # kernels whose instruction lines the parser no longer recognises
    odd.write_text("0000000000001000 <kernel_a>:\n  s_endpgm ; 1000: BF810000\n")
    for path in (str(empty), str(odd), "/bin/ls", str(tmp_path / "missing.so")):
        rc, out = _run(path)
        assert rc == 2, (path, out)
        with pytest.raises(RuntimeError, match="NOT CHECKED"):
            b.check_exec_zero(path)


def test_matrix_libraries_carry_their_verdict():
    """Where the launch-bounds matrix has been built (scripts/build_clip_matrix.py), every library has a recorded verdict, the verdict is
    what the checker says now, and the two-wave ... four-wave builds of the shipped sources are listed for the GPU tier to use."""
    from fastx_toolkit_amd import build as b
    found = 0
    for w in (2, 3, 4):
        so = os.path.join(b.PKG, "libfxg_m_w%d.so" % w)
        if not os.path.exists(so):
            continue
        found += 1
        v = json.load(open(so[:-3] + ".json"))
        rc, out = _run(so)
        assert v["waves"] == w and v["accepted"] == (rc == 0), (w, out[-600:])
    if not found:
        pytest.skip("matrix libraries not built here (scripts/build_clip_matrix.py); the GPU tier builds them")
