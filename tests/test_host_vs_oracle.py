"""Host logic of the product (phases 1-3 in C++, gaast_amd/csrc/host) against the oracle.

"bit-exact for grade/index bookkeeping": every node kind, child link, maximal / minimal grade
set, use count and every comp-mul entry (indices and f64 coefficient) must be identical.
No GPU needed.
"""
import numpy as np
import pytest

import gaast_amd as ga
from exprs import CASES
from helpers import HipBackend, OracleBackend, assert_same_ast
from oracle import pyoracle as og


@pytest.mark.parametrize("name", sorted(CASES))
def test_specialized_ast_matches_oracle(name):
    alg, build = CASES[name]
    o = build(OracleBackend(), np.random.default_rng(7)).specialize(alg)
    h = build(HipBackend(), np.random.default_rng(7)).specialize(alg, materialize_limit=0)
    assert_same_ast(o, h)


def test_compact_descriptor_counts_match():
    """A product kept as a compact descriptor reports the exact length its list would have."""
    alg, build = CASES["r6_gp_full"]
    h = build(HipBackend(), np.random.default_rng(7)).specialize(alg, materialize_limit=10)
    n = h.get_node(h.root_id())
    assert n.n_comp_muls == 4 ** 6 and h.comp_muls(h.root_id()) is None
    alg, build = CASES["r6_gp_even"]
    o = build(OracleBackend(), np.random.default_rng(7)).specialize(alg)
    h = build(HipBackend(), np.random.default_rng(7)).specialize(alg, materialize_limit=10)
    assert o.nodes()[o.root()].n_comp_muls == h.get_node(h.root_id()).n_comp_muls


def test_blade_index_rules_match_oracle():
    """T2/T3: component <-> blade maps, reordering sign and metric coefficient, n <= 8."""
    L, G = og.lib(), ga.lib()
    import ctypes as C
    for n in range(0, 9):
        for k in range(n + 1):
            for i in range(L.og_n_choose_k(n, k)):
                b = L.og_index_to_bitfield_permut(n, k, i)
                assert G.gaast_component_to_blade(n, k, i) == b
                g = C.c_int()
                assert G.gaast_blade_to_component(n, b, C.byref(g)) == i and g.value == k
    rng = np.random.default_rng(3)
    for diag in ([1.0] * 6, [1.0, 1.0, 1.0, 1.0, -1.0], [0.0, 1.0, 1.0], [2.0, -0.5, 3.0, 0.25, -1.5]):
        n = len(diag)
        oa = og.diag_metric(diag)
        cd = (C.c_double * n)(*diag)
        for _ in range(400):
            a, b = int(rng.integers(0, 1 << n)), int(rng.integers(0, 1 << n))
            r1, r2 = C.c_uint64(), C.c_uint64()
            c1 = L.og_ortho_basis_blades_gp(C.byref(oa), a, b, C.byref(r1))
            c2 = G.gaast_blades_gp(n, cd, a, b, C.byref(r2))
            assert r1.value == r2.value and c1 == c2 and np.signbit(c1) == np.signbit(c2)


def test_grade_set_ops_match_oracle():
    rng = np.random.default_rng(5)
    for _ in range(300):
        a, b = int(rng.integers(0, 1 << 9)), int(rng.integers(0, 1 << 9))
        oa, ob = og.GradeSet.from_mask(a), og.GradeSet.from_mask(b)
        ha, hb = ga.GradeSet(a), ga.GradeSet(b)
        assert (oa * ob).mask == (ha * hb).mask
        assert (oa + ob).mask == (ha + hb).mask
        assert oa.intersection(ob).mask == ha.intersection(hb).mask
        s = int(rng.integers(0, 1 << 12))
        for kind in range(5):
            ol, orr = og.GradeSet.from_mask(s).parts_contributing_to_product(kind, oa, ob)
            hl, hr = ga.GradeSet(s).parts_contributing_to_product(kind, ha, hb)
            assert (ol.mask, orr.mask) == (hl.mask, hr.mask)


def test_reference_panics_are_reported_not_raised_as_crashes():
    e1, e2, e3 = ga.Expr.basis_vectors(3)
    with pytest.raises(ga.GaastError):
        (e1 + (e1 ^ e2)).exp().specialize(3)     # exp of a non-k-vector: grade_set.rs:182-185
    with pytest.raises(og.OraclePanic):
        o1, o2, _ = og.Expr.basis_vectors(3)
        (o1 + (o1 ^ o2)).exp().specialize(3)


def test_custom_product_selection():
    """Expr::product with a user closure (expr.rs:123-144): here the scalar product."""
    rng = np.random.default_rng(11)
    from helpers import random_mv, full_grades
    a, b = random_mv(rng, 4, full_grades(4)), random_mv(rng, 4, full_grades(4))
    h = ga.mv(ga.GradeMapMV(a, 4)).product(ga.mv(ga.GradeMapMV(b, 4)), lambda k1, k2: [0]).specialize(4)
    root = h.get_node(h.root_id())
    assert root.minimal_grade_mask == 1 and root.n_comp_muls == 16
    assert all(m[0] == m[2] and m[4] == 0 for m in h.comp_muls(h.root_id()))
