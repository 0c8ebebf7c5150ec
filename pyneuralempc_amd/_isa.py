"""Build stage between hipcc's code generation and the assembler: find, and repair, vector instructions that the compiler
placed in front of an exec restore (gfx950 assembly text in, gfx950 assembly text out).

The defect (AMD clang 22 / ROCm 7.2, found in round 4 from the wrong rows of three wave-per-tile kernels in round 3):
when the register allocator needs a vector-register spill store (`v_accvgpr_write_b32 aN, vM`, `scratch_store_*`), a reload
or a live-range-split copy at the top of the block that JOINS a divergent region, it can put it in front of the
`s_or_b64 exec, exec, s[..]` that re-opens the lanes the region had closed:

    s_and_saveexec_b64 s[0:1], s[12:13]      ; lanes with e < 16 * ne  (none at all when ne == 0)
    s_cbranch_execz .LBB47_49
      ...                                    ; the region's body
  .LBB47_49:
    v_accvgpr_write_b32 a163, v18            ; <-- spill store of a value EVERY lane needs later, executed by the region's
    s_or_b64 exec, exec, s[0:1]              ;     lanes only: with exec == 0 nothing is stored at all
      ...
    v_accvgpr_read_b32 v18, a163             ; reload, all lanes: whatever the register file held (machine dependent)

`v_readlane / v_writelane / v_readfirstlane` and scalar instructions do not depend on exec and are what the compiler means to
have there (scalar-register spill reloads).  Moving the stray spill stores directly behind the restore is the placement the
allocator should have chosen.  `repair` moves ONLY spill stores (`v_accvgpr_write_b32 aN, vM`, `scratch_store_* ... Folded
Spill`) and only when what they hop over is independent of them (`_check_movable`); every other finding stops the build
(`IsaRepairError`).  The same placement is looked for at the other two places lanes come back: the else entry of an
if / else (`s_or_saveexec_b64`) and the restore behind a divergent loop, reached by falling through its `s_cbranch_execnz`
with exec == 0.  `tools/isa_defect_repro.hip` is a standalone reproducer of the compiler's placement, compiled and scanned by
`tests/test_build_isa_cpu.py`: after a toolchain bump that test says whether the defect is still there.

A second check, same stage (`scan_store_hazard`): a store of more than 64 bits written as inline asm whose data registers a
vector instruction overwrites within two wait states.  The compiler pads its own wide stores (gfx9 "store data hazard") but
cannot see into an asm statement; round 4's first sparse-contract epilogue had `v_lshl_add_u64 v[2:3], ...` directly behind
`global_store_dwordx4 v[12:13], v[2:5]` and lanes 12..15 of every 16 stored garbage.  Such stores carry their own `s_nop 1`
in the source; the check keeps it that way.

`scan` is also the check of the finished code: `_build.py` runs it on the repaired text and fails the build on any finding;
`tests/test_cabi_cpu.py` runs both on text fixtures and on the reports the build leaves next to the objects.
"""
import re

LABEL = re.compile(r"^([.\w$]+):")
KERNEL = re.compile(r"^(_Z\w+):")
SKIP = re.compile(r"^\s*s_cbranch_execz\s+([.\w$]+)")
LOOP_BACK = re.compile(r"^\s*s_cbranch_execnz\b")
# the instructions that give lanes back: the plain restore at a join, and the else entry of an if / else (`s_or_saveexec_b64`
# at the top of the flow block: exec |= the lanes that skipped the `then` side)
EXEC_RESTORE = re.compile(r"^\s*(s_or_b64\s+exec,\s*exec,\s*s\[|s_or_saveexec_b64\b)")
# what may stand in front of a restore: scalar instructions and the lane-addressed moves (exec-independent)
SAFE = re.compile(r"^\s*(s_\w+|v_readlane_b32|v_writelane_b32|v_readfirstlane_b32)\b")
BRANCH = re.compile(r"^\s*s_(cbranch\w*|branch|endpgm|setpc_b64)\b")
# anything else that writes exec opens a region of its own (an `if` without a skip branch: saveexec, body, restore in one
# block): what stands between it and its restore is the region's body, not a misplaced instruction
EXEC_WRITE = re.compile(r"^\s*(s_\w+saveexec_b64\b|s_\w+\s+exec\s*,|v_cmpx_)")
# the only instructions `repair` moves: the register allocator's spill stores.  Anything else in that position is not this
# defect and fails the build (a lane-masked copy a later compiler puts there on purpose must not be moved silently)
SPILL_AGPR = re.compile(r"^\s*v_accvgpr_write_b32\s+a(\d+),\s*v(\d+)\s*(;.*)?$")
SPILL_SCRATCH = re.compile(r"^\s*scratch_store_\w+\s+.*;\s*\d+-byte Folded Spill\s*$")
_SREG = re.compile(r"\bs(\d+)\b|\bs\[(\d+):(\d+)\]")
_VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


class IsaRepairError(RuntimeError):
    """the text holds the defect in a form `repair` does not move: the build stops"""


def _is_code(text):
    s = text.strip()
    return bool(s) and not s.startswith(";") and not s.startswith(".")


def _regs(rx, text):
    """register numbers of one file (s or v) an instruction mentions, ranges expanded; the comment is not an operand"""
    out = set()
    for m in rx.finditer(text.split(";")[0]):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def _walk(lines):
    """Yields (kernel, restore_index, [indices of vector instructions that run in front of the restore with the lanes of
    the region that ends there -- or with none].  Three placements:
      * a join label (target of `s_cbranch_execz`) ... `s_or_b64 exec, exec, s[..]`          the skipped `if`
      * a join label ... `s_or_saveexec_b64`                                                 the else entry of an if / else
      * the fall-through of a loop's `s_cbranch_execnz` ... either restore                   a divergent loop's exit: exec == 0
    A label in between does not end the stretch (the lanes that arrived at the join flow through it, too)."""
    joins = set(m.group(1) for m in (SKIP.match(t) for t in lines) if m)
    kernel = None
    block = None            # indices since the last join label / loop exit; None = exec is not known to be narrowed here
    for i, text in enumerate(lines):
        m = KERNEL.match(text)
        if m:
            kernel, block = m.group(1), None
            continue
        m = LABEL.match(text)
        if m:
            if m.group(1) in joins:
                block = []
            continue
        if not _is_code(text):
            continue
        if EXEC_RESTORE.match(text):
            bad = [j for j in (block or []) if not SAFE.match(lines[j])]
            if bad:
                yield kernel, i, bad
            block = None
            continue
        if LOOP_BACK.match(text):
            block = []                      # not taken: every lane has left the loop, exec == 0 until the restore
            continue
        if BRANCH.match(text) or EXEC_WRITE.match(text):
            block = None
            continue
        if block is not None:
            block.append(i)


def scan(text):
    """-> [{"kernel", "line" (1-based, of the restore), "instructions": [(line, text), ...]}, ...]"""
    lines = text.split("\n")
    return [{"kernel": k, "line": r + 1, "instructions": [(j + 1, lines[j].strip()) for j in bad]}
            for k, r, bad in _walk(lines)]


def _check_movable(lines, kernel, r, bad):
    """The moved instructions hop over whatever stays between the first of them and the restore (scalar instructions,
    lane-addressed moves).  That is only the same program when (i) every moved instruction is a spill store, (ii) no wait
    count stands in between (a moved memory instruction would leave its cover), (iii) nothing in between writes or reads a
    register the moved instructions use or define: a `v_writelane_b32 vN` in front of the moved spill of vN, an `s_mov` of
    a scratch store's base."""
    where = f"{kernel}, line {r + 1}"
    for j in bad:
        t = lines[j]
        if not (SPILL_AGPR.match(t) or SPILL_SCRATCH.match(t)):
            raise IsaRepairError(f"{where}: `{t.strip()}` stands in front of an exec restore and is not a register-allocator "
                                 "spill store (v_accvgpr_write_b32 aN, vM / scratch_store_* ; Folded Spill): not moved")
    mv_v = set().union(*(_regs(_VREG, lines[j]) for j in bad))
    mv_s = set().union(*(_regs(_SREG, lines[j]) for j in bad))
    has_mem = any(SPILL_SCRATCH.match(lines[j]) for j in bad)
    badset = set(bad)
    for j in range(bad[0] + 1, r):
        t = lines[j]
        if j in badset or not _is_code(t):
            continue
        name = t.split()[0]
        if name == "s_waitcnt" and has_mem:
            raise IsaRepairError(f"{where}: a spill store to scratch would be moved across `{t.strip()}`")
        if _regs(_VREG, t) & mv_v or _regs(_SREG, t) & mv_s:
            raise IsaRepairError(f"{where}: `{t.strip()}` between the spill store and the restore touches a register the "
                                 "moved instruction uses")


def repair(text):
    """Moves every stray spill store directly behind its restore.  -> (new text, findings of the input).  Raises
    IsaRepairError on a finding that is not a spill store or that cannot be moved without changing the program."""
    lines = text.split("\n")
    found = list(_walk(lines))
    findings = [{"kernel": k, "line": r + 1, "instructions": [(j + 1, lines[j].strip()) for j in bad]} for k, r, bad in found]
    for k, r, bad in found:
        _check_movable(lines, k, r, bad)
    # back to front, so that earlier indices stay valid
    for _, r, bad in reversed(found):
        moved = [lines[j] for j in bad]
        for j in reversed(bad):
            del lines[j]
        r -= len(bad)
        lines[r + 1:r + 1] = moved
    return "\n".join(lines), findings


# ---- inline-asm wide stores: the store-data hazard the compiler cannot pad for (module docstring)
_ASM_START, _ASM_END = re.compile(r"^\s*;;#ASMSTART"), re.compile(r"^\s*;;#ASMEND")
_WIDE_STORE = re.compile(r"^\s*(?:global|flat|scratch)_store_dwordx[34]\s+[^,]+,\s*v\[(\d+):(\d+)\]")
_VALU_DEF = re.compile(r"^\s*v_\w+\s+(?:v(\d+)\b|v\[(\d+):(\d+)\])")
_S_NOP = re.compile(r"^\s*s_nop\s+(\d+)")
STORE_DATA_WAIT_STATES = 2       # gfx940+ (LLVM GCNHazardRecognizer: VALU write of >64-bit VMEM store data)


def scan_store_hazard(text):
    """-> [{"kernel", "line", "store", "clobber"}]: asm-block stores of > 64 bits with a VALU write of their data registers
    inside the hazard window."""
    lines = text.split("\n")
    out, kernel, in_asm = [], None, False
    for i, t in enumerate(lines):
        m = KERNEL.match(t)
        if m:
            kernel = m.group(1)
        if _ASM_START.match(t):
            in_asm = True
            continue
        if _ASM_END.match(t):
            in_asm = False
            continue
        m = _WIDE_STORE.match(t) if in_asm else None
        if not m:
            continue
        lo, hi = int(m.group(1)), int(m.group(2))
        waited = 0
        for j in range(i + 1, min(i + 40, len(lines))):
            u = lines[j]
            if not _is_code(u) or _ASM_START.match(u) or _ASM_END.match(u):
                continue
            if LABEL.match(u) or BRANCH.match(u):
                break
            n = _S_NOP.match(u)
            if n:
                waited += int(n.group(1)) + 1
            else:
                d = _VALU_DEF.match(u)
                if d and not u.strip().startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
                    a = int(d.group(1)) if d.group(1) is not None else int(d.group(2))
                    b = int(d.group(1)) if d.group(1) is not None else int(d.group(3))
                    if a <= hi and b >= lo:
                        out.append({"kernel": kernel, "line": i + 1, "store": t.strip(), "clobber": u.strip()})
                        break
                waited += 1
            if waited >= STORE_DATA_WAIT_STATES:
                break
    return out
