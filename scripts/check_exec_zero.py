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

This script disassembles every kernel of the library, builds its control flow graph and follows what is known about EXEC: it is zero
on the fall-through edge of `s_cbranch_execnz` (a lane-divergent loop's back edge: falling through means no lane is left) and on the
taken edge of `s_cbranch_execz`, until something writes it.  Reported are
  (a) every vector / memory instruction in a block that is ONLY ever entered with EXEC = 0, ahead of the block's EXEC restore (the
      listing above), and
  (b) scratch stores ahead of the EXEC restore of a block that CAN be entered with EXEC = 0 (the same misplacement in a join block).
v_readlane / v_readfirstlane / v_writelane ignore EXEC and are not reported, nor is the bracket `s_or_saveexec_b64 sN, -1 ... s_mov_b64
exec, sN` (whole-wave spill of an SGPR-spill register, which sets the mask itself).

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


def code_objects(so, tmp):
    """The gfx950 code objects embedded in a HIP shared library -- one per translation unit (the engine is built from eight, csrc/fxg_host.h) -- or the
    file itself if it already is one."""
    out = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-h", so], capture_output=True, text=True).stdout
    if "AMDGPU" in out or "amdgpu" in out:
        return [so]
    work = os.path.join(tmp, os.path.basename(so))
    shutil.copy(so, work)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", work], capture_output=True, text=True, cwd=tmp)
    found = [os.path.join(tmp, f) for f in sorted(os.listdir(tmp)) if f.startswith(os.path.basename(so) + ".") and "amdgcn" in f]
    if not found:
        raise RuntimeError("no gfx950 code object found in %s" % so)
    return found


def check_kernel(name, ins, full_lines, joins=False):
    """Every vector / memory instruction of one kernel that is CERTAIN to run with EXEC = 0, by a forward data flow over the kernel's
    control flow graph.  EXEC is known to be zero on the fall-through edge of `s_cbranch_execnz` and on the taken edge of `s_cbranch_execz`;
    a block starts with EXEC = 0 when every edge into it carries that; any write of EXEC ends the knowledge (the bracket
    `s_or_saveexec_b64 sN, -1 ... s_mov_b64 exec, sN` of a whole-wave spill sets the mask itself and then restores what it found).
    Returns [(address of the first such instruction of a block, [instruction texts])]."""
    if not ins:
        return []
    base = ins[0][0]
    addr_index = {a: i for i, (a, _, _, _) in enumerate(ins)}
    tgt_of = {}
    for i, (a, mn, ops, _) in enumerate(ins):
        if mn.startswith("s_cbranch") or mn == "s_branch":
            m = re.search(r"\+0x([0-9a-f]+)>", full_lines[i])
            t = base + int(m.group(1), 16) if m else (base if re.search(r"<[^>+]+>\s*$", full_lines[i]) else None)
            if t in addr_index:
                tgt_of[i] = addr_index[t]
    leaders = {0} | set(tgt_of.values()) | {i + 1 for i in tgt_of if i + 1 < len(ins)}
    leaders = sorted(leaders)
    block_of = {}
    for b, l in enumerate(leaders):
        for k in range(l, leaders[b + 1] if b + 1 < len(leaders) else len(ins)):
            block_of[k] = b
    nb = len(leaders)
    ends = [(leaders[b + 1] if b + 1 < nb else len(ins)) - 1 for b in range(nb)]

    def writes_exec(mn, ops):
        return mn.startswith("s_") and (re.match(r"^(exec|exec_lo|exec_hi)\b", ops) is not None or "saveexec" in mn)

    def walk(b, zero, report=None):
        """state at the end of block b when it is entered with EXEC = 0 (zero) or unknown; collects the instructions that run under 0"""
        wwm = None                                            # (register, state before the bracket)
        for k in range(leaders[b], ends[b] + 1):
            a, mn, ops, txt = ins[k]
            if mn.startswith("s_or_saveexec") and ops.replace(" ", "").endswith(",-1"):
                wwm = (ops.split(",")[0].strip(), zero)
                zero = False
                continue
            if wwm and mn == "s_mov_b64" and ops.replace(" ", "") == "exec,%s" % wwm[0]:
                zero, wwm = wwm[1], None
                continue
            if writes_exec(mn, ops):
                zero, wwm = False, None
                continue
            if zero and report is not None and ((mn.startswith("v_") and mn not in IGNORES_EXEC) or mn.startswith(MEM_PREFIX)):
                report.append((a, txt))
        return zero

    # edges: (from block, to block, state override) -- override True: EXEC = 0 on this edge whatever the block ends with, False: not zero
    preds = [[] for _ in range(nb)]
    for b in range(nb):
        k = ends[b]
        a, mn, ops, _ = ins[k]
        nxt = block_of.get(k + 1)
        if mn == "s_branch":
            if k in tgt_of:
                preds[block_of[tgt_of[k]]].append((b, None))
        elif mn.startswith("s_cbranch"):
            if k in tgt_of:
                preds[block_of[tgt_of[k]]].append((b, True if mn == "s_cbranch_execz" else False if mn == "s_cbranch_execnz" else None))
            if nxt is not None:
                preds[nxt].append((b, True if mn == "s_cbranch_execnz" else False if mn == "s_cbranch_execz" else None))
        elif mn not in ("s_endpgm", "s_setpc_b64") and nxt is not None:
            preds[nxt].append((b, None))
    entry = [True] * nb                                       # optimistic start; the kernel's first block is entered with live lanes
    entry[0] = False
    changed = True
    while changed:
        changed = False
        out = [walk(b, entry[b]) for b in range(nb)]
        for b in range(1, nb):
            if not entry[b]:
                continue
            ok = bool(preds[b]) and all((ov if ov is not None else out[p]) for p, ov in preds[b])
            if not ok:
                entry[b] = False
                changed = True
    found = []
    for b in range(nb):
        if entry[b]:
            rep = []
            walk(b, True, rep)
            if rep:
                found.append((rep[0][0], [t for _, t in rep]))
    # Second form: a block that CAN be entered with EXEC = 0 (the taken edge of a skip branch, or a loop exit, is among its edges) but also
    # with some lanes live, and that has scratch STORES ahead of its EXEC restore.  On the zero path the store is lost, on the others it
    # covers only the lanes of that path; the lanes the restore brings back reload what the slot held before -- from an earlier launch
    # if nothing else wrote it.  (Found as a 48-column instance at four waves per SIMD that was wrong on the FIRST launch after another
    # kernel had used the scratch and right on every repetition: the stale slot then held the same launch's own values.)  hipcc's regular
    # output has no such block: four builds of the 62 kernels show exactly that one.
    out = [walk(b, entry[b]) for b in range(nb)]
    for b in range(1, nb):
        if entry[b] or not any((ov if ov is not None else out[p]) for p, ov in preds[b]):
            continue
        rep = []
        walk(b, True, rep)
        rep = [(a, t) for a, t in rep if t.startswith("scratch_store")]
        if rep:
            found.append((-rep[0][0], [t for _, t in rep]))
    return found


def check(so, joins=False):
    if so.endswith(".dis"):                                   # a saved `llvm-objdump -d` listing (tests/golden/exec_zero_*.dis)
        text = open(so).read()
    else:
        with tempfile.TemporaryDirectory() as tmp:
            text = "\n".join(subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], capture_output=True, text=True, check=True).stdout
                             for co in code_objects(os.path.abspath(so), tmp))
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
    check.last_instructions = sum(len(ins) for ins, _ in ks.values())
    return report, len(ks)


# exit codes: 0 = every kernel read and none has the pattern; 1 = the pattern was found (a real finding); 2 = the check could not do its
# job -- the tools are missing, the code object could not be taken out, or the listing did not parse into kernels with instructions (a
# changed llvm-objdump format must not pass as "0 kernels, 0 places")
def main(paths):
    bad = 0
    for so in paths:
        try:
            report, nk = check(so)
            ni = check.last_instructions
        except Exception as e:                                     # noqa: BLE001 -- any tool failure is "could not check", never "clean"
            print("%s: the ISA check could not run: %s: %s" % (os.path.basename(so), type(e).__name__, e))
            return 2
        if nk == 0 or ni < 8 * nk:
            print("%s: the ISA check read %d kernels with %d instructions -- the disassembly was not understood" % (os.path.basename(so), nk, ni))
            return 2
        for name, addr, dead in report:
            bad += 1
            mem = [d for d in dead if d.startswith(MEM_PREFIX)]
            where = "block at 0x%X is only ever entered with EXEC = 0" % addr if addr > 0 else "block at 0x%X can be entered with EXEC = 0 and spills ahead of its EXEC restore" % -addr
            print("%s: %s: %s: %d vector/memory instructions ahead of the EXEC restore (%d of them memory), e.g.\n      %s" % (
                os.path.basename(so), name, where, len(dead), len(mem), "\n      ".join(dead[:4])))
        print("%s: %d kernels, %d places with vector/memory code ahead of the EXEC restore" % (os.path.basename(so), nk, len(report)))
    return 1 if bad else 0


check.last_instructions = 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
