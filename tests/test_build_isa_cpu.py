"""CPU: the build's middle stage (pyneuralempc_amd/_isa.py) -- the check for vector instructions in front of an exec restore
and their repair -- on text fixtures, and on the reports the build leaves next to the objects of the shipped library."""
import os

from conftest import REPO

# the shape of the round-3 failure (rows_mfma_kernel<double, 128, 2, false, 4, relu>): the spill store of v18 runs with the
# lanes of the skipped region (none), the reload later reads what the register file held
BROKEN = """\
_ZN5nempc16rows_mfma_kernelIdLi128ELi2ELb0ELi4ELi2EEEvNS_10MfmaParamsE:
\ts_and_saveexec_b64 s[0:1], s[12:13]
\ts_cbranch_execz .LBB47_49
; %bb.45:
\tv_accvgpr_read_b32 v8, a130
\tds_write_b64 v3, v[10:11]
.LBB47_49:                              ;   in Loop: Header=BB47_3 Depth=1
\tv_readlane_b32 s4, v255, 3
\tv_accvgpr_write_b32 a163, v18
\tscratch_store_dword off, v7, off offset:12 ; 4-byte Folded Spill
\ts_or_b64 exec, exec, s[0:1]
\t; wave barrier
\tv_accvgpr_read_b32 v18, a163
\ts_endpgm
"""

# three shapes that are fine: a body laid out of line with its own copy of the restore (entered by execnz), an `if` without
# a skip branch (saveexec, body, restore in one block), and a join block whose prologue is scalar / lane-addressed only
FINE = """\
_ZN5nempc4okayEv:
\ts_and_saveexec_b64 s[0:1], s[2:3]
\ts_cbranch_execnz .LBB1_218
.LBB1_204:
\ts_or_b64 exec, exec, s[0:1]
\ts_endpgm
.LBB1_218:
\ts_nop 14
\tds_write_b64 v252, a[8:9]
\ts_or_b64 exec, exec, s[0:1]
\ts_and_saveexec_b64 s[0:1], vcc
\tv_mov_b32_e32 v1, 0
\tglobal_store_dword v[2:3], v1, off
\ts_or_b64 exec, exec, s[0:1]
\ts_and_saveexec_b64 s[4:5], vcc
\ts_cbranch_execz .LBB1_9
\tv_add_u32_e32 v4, 64, v4
.LBB1_9:
\tv_readlane_b32 s12, v255, 23
\ts_waitcnt vmcnt(0)
\ts_or_b64 exec, exec, s[4:5]
\tv_accvgpr_write_b32 a3, v4
\ts_endpgm
"""


def test_scan_finds_the_vector_instructions_in_front_of_an_exec_restore():
    from pyneuralempc_amd import _isa
    found = _isa.scan(BROKEN)
    assert len(found) == 1
    assert found[0]["kernel"].startswith("_ZN5nempc16rows_mfma_kernel")
    assert [t for _, t in found[0]["instructions"]] == ["v_accvgpr_write_b32 a163, v18",
                                                         "scratch_store_dword off, v7, off offset:12 ; 4-byte Folded Spill"]
    assert _isa.scan(FINE) == []


def test_repair_moves_them_behind_the_restore_and_nothing_else():
    from pyneuralempc_amd import _isa
    fixed, found = _isa.repair(BROKEN)
    assert len(found) == 1 and _isa.scan(fixed) == []
    a, b = BROKEN.split("\n"), fixed.split("\n")
    assert sorted(a) == sorted(b)                                   # the same instructions ...
    i = b.index("\ts_or_b64 exec, exec, s[0:1]")
    assert b[i - 1] == "\tv_readlane_b32 s4, v255, 3"               # ... the lane-addressed move stays in the prologue,
    assert b[i + 1] == "\tv_accvgpr_write_b32 a163, v18"            # the spill stores follow the restore in their order
    assert b[i + 2].startswith("\tscratch_store_dword")
    assert b[i + 3] == "\t; wave barrier"
    same, none = _isa.repair(FINE)
    assert same == FINE and none == []


HAZARD = """\
_ZN5nempc3badEv:
\t;;#ASMSTART
\tglobal_store_dwordx4 v[12:13], v[2:5], off sc0 sc1
\t;;#ASMEND
\tv_lshl_add_u64 v[2:3], v[8:9], 0, 48
\ts_endpgm
_ZN5nempc4goodEv:
\t;;#ASMSTART
\tglobal_store_dwordx4 v[12:13], v[2:5], off sc0 sc1
\ts_nop 1
\t;;#ASMEND
\tv_lshl_add_u64 v[2:3], v[8:9], 0, 48
\t;;#ASMSTART
\tglobal_store_dwordx4 v[12:13], v[2:5], off sc0 sc1
\t;;#ASMEND
\tv_add_u32_e32 v9, 1, v9
\tv_mov_b32_e32 v6, 0
\tv_mov_b32_e32 v2, 0
\tglobal_store_dwordx4 v[12:13], v[2:5], off
\tv_mov_b32_e32 v2, 0
\ts_endpgm
"""


def test_store_data_hazard_of_inline_asm_stores_is_found():
    """A vector write of the data registers of a > 64-bit store inside two wait states: the compiler pads its own stores, an
    asm statement has to carry its `s_nop 1` (round 4: lanes 12..15 of every 16 stored a pointer in the sparse contract)."""
    from pyneuralempc_amd import _isa
    h = _isa.scan_store_hazard(HAZARD)
    assert len(h) == 1 and h[0]["kernel"] == "_ZN5nempc3badEv" and h[0]["clobber"].startswith("v_lshl_add_u64 v[2:3]")


def test_every_translation_unit_of_the_shipped_library_scans_clean():
    """The build fails when a repaired text still has a finding; its reports say what was repaired.  (Round 4: ten join
    blocks in seven units, all in streamed `rows_mfma_kernel` instantiations -- DESIGN.md.)"""
    from pyneuralempc_amd import _build
    reports = {r["unit"]: r for r in _build.isa_reports()}
    lib = os.path.join(REPO, "pyneuralempc_amd", "libnempc.so")
    assert os.path.exists(lib)
    assert sorted(reports) == sorted(_build.SOURCES), "build/<unit>.isa.json missing: rebuild with python -m pyneuralempc_amd._build"
    for unit, rep in reports.items():
        assert rep["left"] == [], unit
        assert rep["store_hazards"] == [], unit
        for f in rep["repaired"]:
            assert f["instructions"], unit
