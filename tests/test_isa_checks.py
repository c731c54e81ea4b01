"""CPU checks of the built device code (no GPU): the scalar-path forward pass's record loads are inline assembly the compiler
does not see, so the ISA is inspected after the build (tools/check_smem_hazard.py) — no instruction may touch the loaded scalar
registers between an s_load_dwordx16 and the s_waitcnt lgkmcnt(0) that ends its flight, along any control-flow path."""
import os
import sys

import pytest

from conftest import PKG, ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_kernel_resources as res  # noqa: E402
import check_smem_hazard as chk  # noqa: E402

HEAD = "0000000000001000 <kern>:\n"


def _line(off, text, target=None):
    tail = " <kern+0x%x>" % target if target is not None else ""
    return "\t%s // %012X: BF800000%s\n" % (text.ljust(50), 0x1000 + off, tail)


def test_checker_on_synthetic_streams():
    ok = HEAD + _line(0, "s_load_dwordx16 s[36:51], s[2:3], 0x0") + _line(8, "v_add_f64 v[0:1], v[2:3], v[4:5]") + \
        _line(16, "s_add_u32 s2, s2, 0x80") + _line(24, "s_waitcnt lgkmcnt(0)") + _line(28, "v_mul_f64 v[0:1], s[36:37], v[2:3]")
    assert chk.check(ok) == ([], 1)
    early = HEAD + _line(0, "s_load_dwordx16 s[36:51], s[2:3], 0x0") + _line(8, "v_mul_f64 v[0:1], s[40:41], v[2:3]") + _line(16, "s_waitcnt lgkmcnt(0)")
    v, n = chk.check(early)
    assert n == 1 and len(v) == 1 and "s[40:41]" in v[0][2]
    spill = HEAD + _line(0, "s_load_dwordx16 s[36:51], s[2:3], 0x0") + _line(8, "v_writelane_b32 v252, s37, 3") + _line(16, "s_waitcnt lgkmcnt(0)")
    assert len(chk.check(spill)[0]) == 1
    # the loop shape of forward_smem: the request at the end of the body, the wait at the top, reached through the back edge
    loop = HEAD + _line(0, "s_waitcnt lgkmcnt(0)") + _line(4, "v_mul_f64 v[0:1], s[36:37], v[2:3]") + \
        _line(12, "s_load_dwordx16 s[36:51], s[2:3], 0x0") + _line(20, "s_cmp_eq_u32 s18, 0") + _line(24, "s_cbranch_scc0 65529", 0) + \
        _line(28, "s_waitcnt lgkmcnt(0)") + _line(32, "s_endpgm")
    assert chk.check(loop) == ([], 1)
    bad_exit = loop.replace("s_waitcnt lgkmcnt(0)".ljust(50) + " // %012X" % (0x1000 + 28), "v_mov_b32_e32 v1, s36".ljust(50) + " // %012X" % (0x1000 + 28))
    assert len(chk.check(bad_exit)[0]) >= 1


def test_built_solve_kernels_have_no_scalar_load_hazard():
    obj = os.path.join(PKG, "build", "cilqr_solve.o")
    if not os.path.exists(obj):
        pytest.skip("build/cilqr_solve.o not present (the library was built elsewhere)")
    violations, loads = chk.check(chk.disassemble(obj))
    assert loads >= 8, "the scalar-path forward pass is missing from the object"
    assert not violations, violations[:3]


def test_register_budget_rule_on_synthetic_notes():
    two = "_ZN5cilqr12_GLOBAL__N_118cilqr_solve_kernelILb0ELi2ELb0ELb0EEEvNS_9SolveArgsE"
    unc = "_ZN5cilqr12_GLOBAL__N_118cilqr_solve_kernelILb0ELi1ELb0ELb1EEEvNS_9SolveArgsE"
    c2 = "_ZN5cilqr12_GLOBAL__N_118cilqr_solve_kernelILb0ELi1ELb0ELb0EEEvNS_9SolveArgsE"
    ok = dict(vgpr=256, agpr=0, vgpr_spill=0, sgpr_spill=60, scratch=0)
    assert res.check({two: dict(ok, vgpr_spill=1, scratch=8), unc: dict(ok, vgpr=274, agpr=18), c2: ok}) == []
    assert len(res.check({two: dict(ok, vgpr=257, agpr=1)})) == 1            # the regression this check exists for
    assert len(res.check({c2: dict(ok, vgpr_spill=3, scratch=24)})) == 1    # the headline kernel must not spill


def test_built_solve_kernels_keep_their_register_budgets():
    obj = os.path.join(PKG, "build", "cilqr_solve.o")
    if not os.path.exists(obj):
        pytest.skip("build/cilqr_solve.o not present (the library was built elsewhere)")
    notes = res.kernel_notes(obj)
    assert sum(1 for k in notes if "cilqr_solve_kernel" in k) >= 24
    assert res.check(notes) == []
