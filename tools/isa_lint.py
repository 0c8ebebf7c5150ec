#!/usr/bin/env python3
"""Static check of gfx950 assembly for one compiler defect: a vector instruction placed between a block label and the
`s_or_b64 exec, exec, s[..]` that re-opens the lanes a divergent region had closed.

hipcc (AMD clang 22, ROCm 7.2) sometimes inserts a vector-register spill (`v_accvgpr_write_b32 aN, vM`, `scratch_store_*`),
its reload, or a live-range-split copy (`v_mov_b32`) at the TOP of the block that ends a divergent region -- in front of the
exec restore.  The instruction then runs with the lanes of the region only (none at all when the region was skipped), the
other lanes of the destination keep whatever the register file held, and a value that every lane needs later is garbage in
those lanes.  That is what made three wave-per-tile row kernels return wrong, machine-dependent rows in round 3 (DESIGN.md
"The exec-restore spill defect").  `v_readlane / v_writelane` and scalar instructions ignore exec and are fine there.

    python tools/isa_lint.py file.s [file.s ...]         # report per kernel, exit status 1 when anything is found
"""
import re
import sys

LABEL = re.compile(r"^([.\w$]+):")
EXEC_RESTORE = re.compile(r"^\s*s_or_b64\s+exec,\s*exec,\s*s\[")
SAFE = re.compile(r"^\s*(s_\w+|v_readlane_b32|v_writelane_b32|v_readfirstlane_b32|;.*|)\s")
SKIP = re.compile(r"^\s*s_cbranch_execz\s+([.\w$]+)")
KERNEL = re.compile(r"^(_Z\w+):")
BRANCH = re.compile(r"^\s*s_(cbranch\w*|branch|endpgm|setpc_b64)\b")
# anything else that writes exec opens a region of its own (an `if` without a skip branch: saveexec, body, restore, all in one
# block): what stands between it and its restore is the body, not a misplaced instruction
EXEC_WRITE = re.compile(r"^\s*(s_\w+saveexec_b64\b|s_\w+\s+exec\s*,|v_cmpx_)")


def scan(path):
    """-> list of (kernel, line_no_of_restore, [(line_no, text) offending instructions])"""
    findings = []
    kernel = None
    block = None        # (line_no, text) since the last join label; None = not directly behind one
    lines = open(path).read().split("\n")
    # join labels: where the lanes that skipped a region (or left a loop) arrive -- the targets of `s_cbranch_execz`.  A
    # region's body laid out of line (entered by `s_cbranch_execnz`) legitimately ends with its own copy of the restore.
    joins = set(m.group(1) for m in (SKIP.match(t) for t in lines) if m)
    if True:
        for no, text in enumerate(lines, 1):
            m = KERNEL.match(text)
            if m:
                kernel = m.group(1)
                block = None
                continue
            m = LABEL.match(text)
            if m:
                block = [] if m.group(1) in joins else None
                continue
            stripped = text.strip()
            if not stripped or stripped.startswith(";") or stripped.startswith("."):
                continue
            if EXEC_RESTORE.match(text):
                bad = [(n, t) for n, t in (block or []) if not SAFE.match(t + " ")]
                if bad:
                    findings.append((kernel, no, bad))
                block = None    # what follows the restore runs with the lanes re-opened
                continue
            if BRANCH.match(text) or EXEC_WRITE.match(text):
                block = None
                continue
            if block is not None:
                block.append((no, stripped))
    return findings


def main(argv):
    total = 0
    for path in argv:
        for kernel, no, bad in scan(path):
            total += 1
            print(f"{path}:{no}: {kernel}")
            for n, t in bad:
                print(f"    {n}: {t}")
    print(f"{total} exec-restore block(s) with vector instructions in front of the restore")
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
