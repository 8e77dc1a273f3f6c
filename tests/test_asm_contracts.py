"""Build-time facts behind the hand-counted `s_waitcnt vmcnt(N)` literals (the r03 memory fault was a
STATIC property of the ISA -- an inline-asm load still in flight when the compiler reused its destination registers -- so it
is pinned on the ISA, not by a race between two GPU processes):

  * tools/check_asm_contracts.py itself, on synthetic kernels: a correct ring passes; touching a destination before its
    wait, a spill inside the ring and scratch in the metadata are reported;
  * the ISA the shipped library was assembled from (csrc/_obj/*.s, kept by -save-temps; csrc/Makefile runs the same check
    before linking): mt_jump_mfma_kernel, every hamming_mfma_kernel instantiation and gemm_dma_kernel have no scratch, no
    spilled VGPRs and no violation of the FIFO rule;
  * the r03 bug, re-created on the real ISA (the wait that owns the ring's registers weakened to vmcnt(5)), is caught.
"""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "movie-recommendation-engine_amd", "csrc")
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_asm_contracts as cac                                                # noqa: E402

META = """
amdhsa.kernels:
  - .agpr_count:     0
    .name:           _Z4ringv
    .private_segment_fixed_size: {scratch}
    .sgpr_spill_count: 0
    .symbol:         _Z4ringv.kd
    .vgpr_count:     40
    .vgpr_spill_count: {spill}
"""


def kernel(body, scratch=0, spill=0):
    return "_Z4ringv:\n" + body + "\n\ts_endpgm\n.Lfunc_end0:\n" + META.format(scratch=scratch, spill=spill)


RING_OK = """
	;;#ASMSTART
	s_waitcnt vmcnt(0)
	;;#ASMEND
	;;#ASMSTART
	global_load_dwordx4 v[10:13], v[2:3], off
	;;#ASMEND
	;;#ASMSTART
	global_load_dwordx4 v[14:17], v[4:5], off
	;;#ASMEND
.LBB0_1:
	;;#ASMSTART
	s_waitcnt vmcnt(1)
	;;#ASMEND
	v_add_u32_e32 v20, v10, v11
	;;#ASMSTART
	global_load_dwordx4 v[10:13], v[10:11], off
	;;#ASMEND
	;;#ASMSTART
	s_waitcnt vmcnt(1)
	;;#ASMEND
	v_add_u32_e32 v20, v14, v20
	;;#ASMSTART
	global_load_dwordx4 v[14:17], v[4:5], off
	;;#ASMEND
	s_cbranch_scc1 .LBB0_1
	;;#ASMSTART
	s_waitcnt vmcnt(0)
	;;#ASMEND
	global_store_dword v[10:11], v20, off
"""


def test_checker_accepts_a_correct_ring():
    msgs, nk, nsim = cac.check_file("x", kernel(RING_OK))
    assert msgs == [] and nk == 1 and nsim == 1


def test_checker_reports_a_destination_touched_before_its_wait():
    # the r03 fault in miniature: the final wait dropped, the epilogue builds a store address in the ring's registers
    bad = RING_OK.replace("\ts_cbranch_scc1 .LBB0_1\n\t;;#ASMSTART\n\ts_waitcnt vmcnt(0)\n\t;;#ASMEND\n", "\ts_cbranch_scc1 .LBB0_1\n")
    assert bad != RING_OK
    msgs, _, _ = cac.check_file("x", kernel(bad))
    assert any("may still be in flight" in m and "global_store_dword" in m for m in msgs), msgs
    # a use one step too early inside the loop (the literal off by one) -- only visible over the back edge
    bad = RING_OK.replace("\ts_waitcnt vmcnt(1)\n\t;;#ASMEND\n\tv_add_u32_e32 v20, v10, v11", "\ts_waitcnt vmcnt(2)\n\t;;#ASMEND\n\tv_add_u32_e32 v20, v10, v11")
    msgs, _, _ = cac.check_file("x", kernel(bad))
    assert any("v_add_u32_e32 v20, v10, v11" in m for m in msgs), msgs


def test_checker_reports_spills():
    spill = RING_OK.replace("\tv_add_u32_e32 v20, v14, v20\n", "\tscratch_store_dword off, v30, off offset:4\n\tv_add_u32_e32 v20, v14, v20\n")
    msgs, _, _ = cac.check_file("x", kernel(spill, scratch=8, spill=1))
    assert any("private_segment_fixed_size = 8" in m for m in msgs)
    assert any("vgpr_spill_count = 1" in m for m in msgs)
    # the counted wait behind the spill now retires one ring request too few: the next use of the ring is flagged, and so is the
    # counted wait that runs with a compiler-generated request in the FIFO
    assert any("compiler-generated request" in m for m in msgs), msgs


@pytest.fixture(scope="module")
def shipped_isa():
    """the .s files the in-tree objects were assembled from (make rebuilds whatever is stale; no GPU needed)"""
    subprocess.check_call(["make", "-s", "-C", CSRC, "_obj/asm_contracts.ok"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    files = {n: os.path.join(CSRC, "_obj", f"{n}-hip-amdgcn-amd-amdhsa-gfx950.s") for n in ("mt19937", "hamming_mfma", "dense_mfma")}
    for f in files.values():
        assert os.path.exists(f), f
    return {n: open(f).read() for n, f in files.items()}


def test_shipped_ring_kernels_hold_their_contracts(shipped_isa):
    simulated = 0
    for n, asm in shipped_isa.items():
        msgs, nk, nsim = cac.check_file(n, asm)
        assert msgs == [], "\n".join(msgs[:10])
        simulated += nsim
        meta = cac.kernel_metadata(asm)
        for k, d in meta.items():
            assert d["private_segment_fixed_size"] == 0 and d["vgpr_spill_count"] == 0, (k, d)
    meta = {}
    for asm in shipped_isa.values():
        meta.update(cac.kernel_metadata(asm))
    names = " ".join(meta)
    # the kernels the contracts are about are really there (and the spilling, never-launched instantiations are gone)
    assert "mt_jump_mfma_kernel" in names and "gemm_dma_kernel" in names
    ham = [k for k in meta if "hamming_mfma_kernel" in k]
    assert len(ham) >= 16 and simulated >= len(ham) + 2
    assert not [k for k in ham if re.search(r"ILi[12]ELi[01]ELi4ELb0E", k)], "KS <= 2 two-workgroup forms spill: must not be built"
    assert not [k for k in ham if "ILi8ELi0ELi32ELb1E" in k], "<8, 0, 32, true> needs scratch at 256 VGPRs: must not be built"
    assert meta[[k for k in meta if "mt_jump_mfma_kernel" in k][0]]["vgpr_count"] <= 256


def test_the_r03_fault_is_caught_on_the_real_isa(shipped_isa):
    """mt19937.hip's final `s_waitcnt vmcnt(0)` owns the A ring's registers; weakened to vmcnt(5) (= nothing of the ring
    awaited, what the r03 build effectively had when the compiler moved the epilogue in front of a bare wait) the
    epilogue's store addresses land in registers with loads in flight."""
    asm = shipped_isa["mt19937"]
    k0 = asm.index("mt_jump_mfma_kernel")
    start = asm.rfind("\n_Z", 0, asm.index(":", k0)) + 1
    end = asm.index(".Lfunc_end", start)
    body = asm[start:end]
    waits = [m.start() for m in re.finditer(r";;#ASMSTART\s*\n\s*s_waitcnt vmcnt\(0\)", body)]
    assert len(waits) >= 2                                   # prologue wait + the one behind the loop
    last = waits[-1]
    doctored = body[:last] + body[last:].replace("s_waitcnt vmcnt(0)", "s_waitcnt vmcnt(5)", 1)
    msgs, _, _ = cac.check_file("doctored", asm[:start] + doctored + asm[end:])
    assert any("mt_jump_mfma_kernel" in m and "may still be in flight" in m for m in msgs), msgs[:5]
