"""The kernels that issue their MFMA operand loads by hand (accum_mfma.hip, contract_mfma.hip: inline-asm global loads,
exact s_waitcnt vmcnt(N)) are compiled to gfx950 ISA and walked by tools/check_asm_loads.py: no instruction may read or
write a vector register while a hand-issued load into it is still outstanding (the compiler cannot see those loads; round 4
lost a loop bound to one -- DESIGN.md section 3, K1m).  Round 5: the walk follows the control flow, back edges included --
a loop body is walked with what its own bottom left outstanding (the shape of every operand ring in these kernels).
Runs without a GPU (hipcc cross-compiles)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
pytestmark = pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="no hipcc")


@pytest.mark.parametrize("source,flags", [("accum_mfma.hip", ()), ("contract_mfma.hip", ()),
                                          # the A/B builds of the LDS form's fetch depth (shipped: 4) and of the register
                                          # forms' rings (tools/history/r4/exact_depth_sweep.sh)
                                          ("accum_mfma.hip", ("-DNGD_LDS_PF=1",)), ("accum_mfma.hip", ("-DNGD_LDS_PF=2",)),
                                          ("accum_mfma.hip", ("-DNGD_EXACT2_DEPTH=2", "-DNGD_EXACT4_DEPTH=1"))])
def test_no_register_is_touched_while_a_hand_issued_load_into_it_is_outstanding(source, flags):
    """every instantiation the library launches (block forms EXACT 0-5, weighted and not, the contraction's RT x PT tiles)
    is in these two sources"""
    import check_asm_loads
    findings, n_loads = check_asm_loads.check(source, flags)
    assert n_loads > 100  # the walk really saw the kernels
    assert not check_asm_loads.check.notes  # ... with every queue of outstanding loads a block is entered with
    assert not findings, findings[:5]


def test_a_hazard_across_a_loop_back_edge_is_seen():
    """tests/asm_hazard/cross_trip.hip: the load issued at the bottom of a trip is consumed at the top of the next before any
    wait -- clean in program order (what the round-4 walk followed), found along the back edge; the variant that waits at
    the head of every trip passes"""
    import check_asm_loads
    findings, n_loads = check_asm_loads.check(os.path.join(ROOT, "tests", "asm_hazard", "cross_trip.hip"))
    assert n_loads == 4
    assert findings and all("k_ringILb0E" in f[0] for f in findings), findings
    assert any("v_add_f64" in f[2] for f in findings)


def test_the_round_4_fault_is_still_found():
    """the prefetching wavefront's lost loop bound (fuzz case 40501, fixed in 8c2bcc2): the source before the fix fails"""
    import check_asm_loads
    import tempfile
    with tempfile.TemporaryDirectory() as t:
        for f in ("ngsdist_amd/csrc/accum_mfma.hip", "ngsdist_amd/csrc/ngd_internal.h", "ngsdist_amd/csrc/ngd_shard.h",
                  "include/ngsdist_amd.h"):
            r = subprocess.run(["git", "-C", ROOT, "show", "8c2bcc2^:" + f], capture_output=True)
            if r.returncode != 0:
                pytest.skip("no git history here (the snapshot on the GPU box has none)")
            os.makedirs(os.path.dirname(os.path.join(t, f)), exist_ok=True)
            open(os.path.join(t, f), "wb").write(r.stdout)
        findings, _ = check_asm_loads.check(os.path.join(t, "ngsdist_amd/csrc/accum_mfma.hip"))
    assert findings and any("k_accum_mfma" in f[0] for f in findings)
