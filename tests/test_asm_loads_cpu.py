"""The kernels that issue their MFMA operand loads by hand (accum_mfma.hip, contract_mfma.hip: inline-asm global loads,
exact s_waitcnt vmcnt(N)) are compiled to gfx950 ISA and walked by tools/check_asm_loads.py: no instruction may read or
write a vector register while a hand-issued load into it is still outstanding (the compiler cannot see those loads; round 4
lost a loop bound to one -- DESIGN.md section 3, K1m).  Runs without a GPU (hipcc cross-compiles)."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="no hipcc")
@pytest.mark.parametrize("source", ["accum_mfma.hip", "contract_mfma.hip"])
def test_no_register_is_touched_while_a_hand_issued_load_into_it_is_outstanding(source):
    import check_asm_loads
    findings, n_loads = check_asm_loads.check(source)
    assert n_loads > 100  # the walk really saw the kernels
    assert not findings, findings[:5]
