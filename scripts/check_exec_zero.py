#!/usr/bin/env python3
"""Static check of the gfx950 code in a built libfxg*.so for the miscompile behind DESIGN.md section 3 "the wrong clip instance".

The pattern (ROCm 7.2's LLVM, found by reading the ISA of the failing builds): a loop whose trip count differs between the lanes
of a wave ends in

        s_andn2_b64 exec, exec, s[88:89]        ; lanes that are done leave the mask
        s_cbranch_execnz .LBB2_479               ; back while any lane is left
    ; %bb.480: %Flow                             ; <- falls through with EXEC = 0
        scratch_store_dwordx2 off, v[192:193], off offset:280 ; 8-byte Folded Spill     <- writes NOTHING
        v_mov_b32_e32 v12, v248                                                         <- copies NOTHING
        ...
        s_or_b64 exec, exec, s[88:89]            ; the lanes come back only here

i.e. under register pressure the allocator puts spill stores and copies of the loop's live-out values into the exit block AHEAD of
the instruction that restores the lanes.  Vector and memory instructions with EXEC = 0 do nothing, so what is reloaded later is
whatever the slot held before the loop.  Nothing in the source is wrong and nothing at run time reports it; which values are hit
changes with every change of the register allocation.

This script disassembles every kernel of the library and reports each vector / memory instruction that is certain to run with
EXEC = 0: the straight-line code between a backward `s_cbranch_execnz` (a lane-divergent loop's back edge: falling through means
no lane is left) and the next write of EXEC.  v_readlane / v_readfirstlane / v_writelane ignore EXEC and are not reported, nor is the
bracket `s_or_saveexec_b64 sN, -1 ... s_mov_b64 exec, sN` (whole-wave spill of an SGPR-spill register, which sets the mask itself).

    python scripts/check_exec_zero.py fastx_toolkit_amd/libfxg.so [more.so ...]     # exit code 1 if anything is found

fastx_toolkit_amd/build.py runs it on every library it builds and refuses the build when it reports anything.
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
IGNORES_EXEC = ("v_readlane_b32", "v_readfirstlane_b32", "v_writelane_b32", "v_nop")
MEM_PREFIX = ("scratch_", "global_", "buffer_", "flat_", "ds_", "tbuffer_")


def code_object(so, tmp):
    """The gfx950 code object embedded in a HIP shared library (or the file itself if it already is one)."""
    out = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-h", so], capture_output=True, text=True).stdout
    if "AMDGPU" in out or "amdgpu" in out:
        return so
    work = os.path.join(tmp, os.path.basename(so))
    shutil.copy(so, work)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", work], capture_output=True, text=True, cwd=tmp)
    for f in sorted(os.listdir(tmp)):
        if f.startswith(os.path.basename(so) + ".") and "amdgcn" in f:
            return os.path.join(tmp, f)
    raise RuntimeError("no gfx950 code object found in %s" % so)


def check_kernel(name, ins, full_lines, joins=False):
    """Returns [(loop branch address, [instruction texts])] for every exit of a lane-divergent loop with vector/memory code ahead of the EXEC restore."""
    base = ins[0][0] if ins else 0
    addr_index = {a: i for i, (a, _, _, _) in enumerate(ins)}
    targets = set()
    tgt_of = {}
    for i, (a, mn, ops, _) in enumerate(ins):
        if mn.startswith("s_cbranch") or mn == "s_branch":
            m = re.search(r"\+0x([0-9a-f]+)>", full_lines[i])
            if m:
                t = base + int(m.group(1), 16)
                targets.add(t)
                tgt_of[i] = t
            elif re.search(r"<[^>+]+>", full_lines[i]):       # branch to the symbol itself
                targets.add(base)
                tgt_of[i] = base
    found = []
    for i, (a, mn, ops, _) in enumerate(ins):
        if mn != "s_cbranch_execnz" or i not in tgt_of or tgt_of[i] > a:
            continue
        dead, wwm = [], None
        for j in range(i + 1, len(ins)):
            aj, mj, oj, tj = ins[j]
            if aj in targets:
                break                                         # another path joins here: its lanes may be live
            if mj.startswith("s_or_saveexec") and oj.endswith("-1"):
                wwm = oj.split(",")[0].strip()
                continue
            if wwm and mj == "s_mov_b64" and oj.replace(" ", "") == "exec,%s" % wwm:
                wwm = None
                continue
            if wwm:
                continue                                      # inside a whole-wave bracket: EXEC = -1
            if (re.match(r"^(exec|exec_lo|exec_hi)\b", oj) and mj.startswith("s_")) or mj.startswith(("s_or_saveexec", "s_and_saveexec", "s_xor_saveexec", "s_andn2_saveexec")):
                break                                         # EXEC is written: the lanes are back
            if mj.startswith("s_cbranch") or mj in ("s_branch", "s_endpgm", "s_setpc_b64"):
                break
            if (mj.startswith("v_") and mj not in IGNORES_EXEC) or mj.startswith(MEM_PREFIX):
                dead.append(tj)
        if dead:
            found.append((a, dead))
    # --joins (investigation only, never part of the gate): `s_cbranch_execz L` is only ever taken with EXEC = 0; vector or memory code
    # between L and the next EXEC restore runs for no lane on the taken path and for the fall-through lanes only on the other.  That is
    # what structured control flow means for a loop latch or a then-block's tail, so most reports are fine; a SPILL STORE there is
    # worth a look (is the value live for the other lanes as well?).
    if not joins:
        return found
    seen = set()
    for i, (a, mn, ops, _) in enumerate(ins):
        if mn != "s_cbranch_execz" or i not in tgt_of or tgt_of[i] in seen or tgt_of[i] not in addr_index:
            continue
        seen.add(tgt_of[i])
        dead, wwm = [], None
        for j in range(addr_index[tgt_of[i]], len(ins)):
            aj, mj, oj, tj = ins[j]
            if j > addr_index[tgt_of[i]] and aj in targets:
                break
            if mj.startswith("s_or_saveexec") and oj.endswith("-1"):
                wwm = oj.split(",")[0].strip()
                continue
            if wwm and mj == "s_mov_b64" and oj.replace(" ", "") == "exec,%s" % wwm:
                wwm = None
                continue
            if wwm:
                continue
            if re.match(r"^(exec|exec_lo|exec_hi)\b", oj) and mj.startswith("s_"):
                break
            if mj.startswith("s_cbranch") or mj in ("s_branch", "s_endpgm", "s_setpc_b64"):
                dead = []                                     # no restore before the next branch: not a join block of this form
                break
            if (mj.startswith("v_") and mj not in IGNORES_EXEC) or mj.startswith(MEM_PREFIX):
                dead.append(tj)
        else:
            dead = []
        if dead:
            found.append((-tgt_of[i], dead))
    return found


def check(so, joins=False):
    if so.endswith(".dis"):                                   # a saved `llvm-objdump -d` listing (tests/golden/exec_zero_*.dis)
        text = open(so).read()
    else:
        with tempfile.TemporaryDirectory() as tmp:
            co = code_object(os.path.abspath(so), tmp)
            text = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], capture_output=True, text=True, check=True).stdout
    report, cur, lines, name = [], None, None, None
    ks = {}
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            name = m.group(1)
            ks[name] = ([], [])
            continue
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*// ([0-9A-F]+):", line)
        if m and name is not None:
            ks[name][0].append((int(m.group(3), 16), m.group(1), m.group(2), line.split("//")[0].strip()))
            ks[name][1].append(line)
    for name, (ins, full) in ks.items():
        for addr, dead in check_kernel(name, ins, full, joins):
            report.append((name, addr, dead))
    return report, len(ks)


def main(paths):
    bad = 0
    joins = "--joins" in paths
    for so in [p for p in paths if p != "--joins"]:
        report, nk = check(so, joins)
        for name, addr, dead in report:
            bad += addr > 0
            mem = [d for d in dead if d.startswith(MEM_PREFIX)]
            where = "exit of the lane-divergent loop at 0x%X" % addr if addr > 0 else "target 0x%X of a skip branch (s_cbranch_execz)" % -addr
            print("%s: %s: %s: %d vector/memory instructions ahead of the EXEC restore (%d of them memory), e.g.\n      %s" % (
                os.path.basename(so), name, where, len(dead), len(mem), "\n      ".join(dead[:4])))
        print("%s: %d kernels, %d places with vector/memory code ahead of the EXEC restore" % (os.path.basename(so), nk, len(report)))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
