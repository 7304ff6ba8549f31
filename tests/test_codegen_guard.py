"""tools/check_spill_exec.py: the static guard against the spill-store placement fault of this toolchain's AMDGPU back end (DESIGN.md
section 10).  Its pattern on two small listings, and the product's device listings (written by __graft_entry__.build()) are clean."""
import glob
import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_spill_exec as G

# the shape found in the Training-mode instantiation of the chain-form build: the join block of an if / else stores the tuple that is
# live across the branch BEFORE it restores EXEC
AFFECTED = """
kern:
	s_and_saveexec_b64 s[0:1], vcc
	s_xor_b64 s[0:1], exec, s[0:1]
	s_cbranch_execz .LBB0_2
; %bb.1:
	v_mul_f64 v[2:3], v[28:29], s[2:3]
.LBB0_2:                             ;   in Loop: Header=BB0_1 Depth=1
	s_andn2_saveexec_b64 s[0:1], s[0:1]
	s_cbranch_execz .LBB0_4
; %bb.3:
	v_cvt_f64_i32_e32 v[28:29], v2
.LBB0_4:                             ;   in Loop: Header=BB0_1 Depth=1
	v_mov_b32_e32 v134, 0x3ff00000
	scratch_store_dwordx4 off, v[150:153], off offset:692 ; 16-byte Folded Spill
	s_or_b64 exec, exec, s[0:1]
	v_add_f64 v[28:29], v[22:23], -v[24:25]
	s_endpgm
"""
# each side of the branch stores its own value of the slot, the join block restores EXEC first: fine
CLEAN = """
kern:
	s_and_saveexec_b64 s[0:1], vcc
	s_cbranch_execz .LBB0_2
; %bb.1:
	scratch_store_dwordx4 off, v[2:5], off offset:136 ; 16-byte Folded Spill
.LBB0_2:
	s_andn2_saveexec_b64 s[0:1], s[0:1]
	s_cbranch_execz .LBB0_4
; %bb.3:
	scratch_store_dwordx4 off, v[28:31], off offset:136 ; 16-byte Folded Spill
.LBB0_4:                             ;   in Loop: Header=BB0_1 Depth=1
	s_or_b64 exec, exec, s[0:1]
	scratch_store_dwordx4 off, v[150:153], off offset:692 ; 16-byte Folded Spill
	scratch_store_dwordx3 v0, v[4:6], off
	s_endpgm
"""


# the second pattern (round 5, the fused Training-mode instantiation after an unrelated change of the Trigger tables): the allocator splits a live
# range around phase B1 — save copy, reuse of the register, copy back — and the SAVE sits at the top of a join block, ahead of its EXEC restore:
# tele_total_time of a finished kart came back as 4.6e-41
AFFECTED_COPY = """
kern:
	s_and_saveexec_b64 s[16:17], vcc
	s_cbranch_execz .LBB29_771
; %bb.769:
	v_div_fixup_f32 v60, v6, v4, v5
.LBB29_771:                             ;   in Loop: Header=BB29_73 Depth=1
	s_or_b64 exec, exec, s[16:17]
.LBB29_772:                             ;   in Loop: Header=BB29_73 Depth=1
	v_mov_b32_e32 v180, v146
	v_mov_b64_e32 v[248:249], v[130:131]
	s_or_b64 exec, exec, s[28:29]
	v_mov_b32_e32 v146, 0x54442d18
	v_mov_b32_e32 v146, v180
	s_endpgm
"""
# a copy AFTER the restore, a constant move before it, and a copy inside a block entered under its own narrowed mask: fine
CLEAN_COPY = """
kern:
	s_and_saveexec_b64 s[16:17], vcc
	s_cbranch_execz .LBB1_2
; %bb.1:
	v_mov_b32_e32 v7, v9
.LBB1_2:
	v_mov_b32_e32 v134, 0x3ff00000
	s_or_b64 exec, exec, s[16:17]
	v_mov_b32_e32 v180, v146
	s_endpgm
"""


def test_second_pattern_register_copies(tmp_path):
    bad = [ins for insts in _kernels(tmp_path, AFFECTED_COPY).values() for _, ins in G.check(insts)]
    assert len(bad) == 2 and "v180, v146" in bad[0] and "v[248:249]" in bad[1]
    assert all(G.check(insts) == [] for insts in _kernels(tmp_path, CLEAN_COPY).values())


def _kernels(tmp_path, text):
    p = tmp_path / "k.s"
    p.write_text(text)
    return G.device_asm(str(p))


def test_pattern(tmp_path):
    bad = [ins for insts in _kernels(tmp_path, AFFECTED).values() for _, ins in G.check(insts)]
    assert len(bad) == 1 and "offset:692" in bad[0]
    assert all(G.check(insts) == [] for insts in _kernels(tmp_path, CLEAN).values())


def test_product_listings_are_clean():
    import __graft_entry__ as ge
    ge.build()                                   # writes build/obj/*.s next to the objects; a flagged unit is rebuilt with the next variant
    listings = sorted(glob.glob(os.path.join(ROOT, "build", "obj", "hk_*.s")))
    if not listings:
        pytest.skip("no device listings here (they do not travel to the GPU box; the record build/obj/codegen_guard.json does)")
    n = 0
    for lst in listings:
        for name, insts in G.device_asm(lst).items():
            assert G.check(insts) == [], (lst, name)
            n += sum("Spill" in i for i in insts)
    assert n > 100          # the tick kernels do spill: the guard had something to look at
