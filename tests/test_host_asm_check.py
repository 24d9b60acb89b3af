"""csrc/asm_check.py: the build-time ISA walk that pins the asm-load / counted-wait scheme of the z-march kernels.
Positive: the gfx950 listings of the shipped kernels pass (hipcc cross-compiles here, no GPU needed).  Negative: listings that reproduce
the failure modes -- a compiler copy of a staged register before the wait (what a spill or a live-range split looks like), the register
handed to another value (the round-4 "3 planes, 4 + 4 waves" fault: an asm-issued load landing in a reused register), a second load into
the same registers, a wave that ends with the load in flight, a path that skips the wait -- are each reported."""
import shutil
from pathlib import Path

import pytest

from pytorch_connectomics_amd.csrc import asm_check

HEAD = "_ZN4pytc4demoEv:\n"
TAIL = "\ts_endpgm\n.Lfunc_end0:\n"


def _k(body: str) -> str:
    return HEAD + body + TAIL


LOAD = "\t;;#ASMSTART\n\tglobal_load_dwordx4 v[4:7], v[2:3], off\n\t;;#ASMEND\n"
WAIT = "\t;;#ASMSTART\n\ts_waitcnt vmcnt(0)\n\t;;#ASMEND\n"


def test_clean_sequences_pass():
    ok = _k(LOAD + "\tv_add_u32_e32 v9, v8, v8\n\tv_mfma_f32_4x4x4_16b_bf16 v[20:23], v[10:11], v[12:13], v[20:23]\n" + WAIT
            + "\tds_write_b16 v30, v4\n")
    assert asm_check.check_listing(ok) == [] and asm_check.asm_loads_in(ok) == 1
    # the address registers of a load may be its neighbour's destination (read at issue): not a violation
    ok2 = _k("\t;;#ASMSTART\n\tglobal_load_dwordx4 v[6:9], v[10:11], off\n\t;;#ASMEND\n\t;;#ASMSTART\n\tglobal_load_dwordx4 v[10:13], v[16:17], off\n\t;;#ASMEND\n"
             + WAIT + "\tds_write_b16 v30, v6\n\tds_write_b16 v30, v10\n")
    assert asm_check.check_listing(ok2) == []
    # a branch on a mask the compiler has just set to a constant has one live edge only
    ok3 = _k(LOAD + "\ts_mov_b64 s[8:9], -1\n\ts_andn2_b64 vcc, exec, s[8:9]\n\ts_cbranch_vccz .LBB0_2\n\tv_mov_b32_e32 v40, v5\n.LBB0_2:\n" + WAIT
             + "\tds_write_b16 v30, v5\n")
    assert asm_check.check_listing(ok3) == []


@pytest.mark.parametrize("name,body,needle", [
    ("copy before the wait (spill / live-range split)", LOAD + "\tv_mov_b32_e32 v40, v5\n" + WAIT, "touches v5"),
    ("scratch spill of a staged register", LOAD + "\tscratch_store_dwordx4 off, v[4:7], off\n" + WAIT, "touches v4, v5, v6, v7"),
    ("register handed to another value", LOAD + "\tv_lshl_add_u64 v[6:7], s[4:5], 0, v[14:15]\n" + WAIT, "touches v6, v7"),
    ("second load into the same registers", LOAD + LOAD + WAIT, "touches v4, v5, v6, v7"),
    ("wave ends with the load in flight", LOAD + "\tv_add_u32_e32 v9, v8, v8\n", "s_endpgm"),
    ("a path around the wait", LOAD + "\ts_cmp_lt_i32 s4, s5\n\ts_cbranch_scc1 .LBB0_3\n" + WAIT + ".LBB0_3:\n\tds_write_b16 v30, v4\n" + WAIT, "touches v4"),
    ("compiler's own wait does not count", LOAD + "\ts_waitcnt vmcnt(0)\n\tv_mov_b32_e32 v40, v5\n" + WAIT, "touches v5"),
])
def test_failure_modes_are_reported(name, body, needle):
    bad = asm_check.check_listing(_k(body))
    assert bad and any(needle in b for b in bad), (name, bad)


@pytest.mark.skipif(shutil.which("hipcc") is None and not Path("/opt/rocm/bin/hipcc").exists(), reason="hipcc not available")
@pytest.mark.parametrize("src", ["dwconv_mfma_kernels.hip", "dwconv_kernels.hip"])
def test_shipped_march_kernels_pass_the_walk(src):
    from pytorch_connectomics_amd.csrc import build
    text = build.asm_listing(build.CSRC / src)
    pats = build.NO_SPILL_KERNELS[src]
    assert asm_check.asm_loads_in(text, pats) >= 50          # fused-block, statistics-only, hi + lo, probe instantiations included
    assert asm_check.check_listing(text, pats) == []
