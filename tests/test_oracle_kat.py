"""Pins the CPU oracle to every known-answer test the reference holds.

Each test mirrors one `#[test]` of the reference, the expected values come from
tests/golden/ref_kat.json (transcribed data).  If these fail the oracle is not trusted and
every parity claim built on it is void.
"""
import json
import os

import numpy as np
import pytest

from oracle import pyoracle as og
from oracle.pyoracle import Expr, GradeSet, grade_map_mv

E = GradeSet.empty
S = GradeSet.single
R = GradeSet.range


@pytest.fixture(scope="module")
def kat(golden_dir):
    with open(os.path.join(golden_dir, "ref_kat.json")) as f:
        return json.load(f)


def _metric(m):
    return 3 if m == "euclid3" else m


def _expected(d):
    return grade_map_mv({int(k): v for k, v in d.items()})


def expr_eq(case, expr):
    """expr_eq! of src/eval.rs:122-128: specialize, eval::<GradeMapMV>(), exact equality."""
    out = expr.specialize(_metric(case["metric"])).eval()
    assert out == _expected(case["expected"]), f"{out} != {case['expected']}"
    # the reference runs its tests in debug builds (overflow checks on): must not panic there
    out_dbg = expr.specialize(_metric(case["metric"])).eval(mode=og.EVAL_DEBUG)
    assert out_dbg == out
    # the oracle's F32 MODE (every operand and result rounded to binary32; the checker of the f32 exact kernels) is pinned by
    # the same four known answers: their inputs, intermediates and results are exactly representable in binary32
    out_f32 = expr.specialize(_metric(case["metric"])).eval(mode=og.EVAL_F32)
    assert out_f32 == out


# ---- src/eval.rs:134-163 -----------------------------------------------------------------
def test_vecs_to_bivec(kat):
    e1, e2, _ = Expr.basis_vectors(3)
    expr_eq(kat["eval"]["vecs_to_bivec"], e1 ^ e2)


def test_vecs_to_trivec(kat):
    e1, e2, e3 = Expr.basis_vectors(3)
    expr_eq(kat["eval"]["vecs_to_trivec"], e2 ^ e1 ^ e3)


def test_vec_norm(kat):
    e0, e1, e2 = Expr.basis_vectors(3)
    expr_eq(kat["eval"]["vec_norm"], (e0 - 2 * e1 + e2).norm_sq())


def test_projection(kat):
    e1, e2, e3 = Expr.basis_vectors(3)
    v = e1 + e2
    bv = 4 * e1 ^ e3
    expr_eq(kat["eval"]["projection"], (v & bv) & bv.vinv())


# ---- src/algebra.rs:274-300 --------------------------------------------------------------
def test_n_choose_k(kat):
    for n, k, want in kat["n_choose_k"]["cases"]:
        assert og.lib().og_n_choose_k(n, k) == want


def test_idx_bitfield_permut_roundtrip(kat):
    n, k = kat["roundtrip"]["index_to_bitfield_to_index"]
    L = og.lib()
    idx = list(range(L.og_n_choose_k(n, k)))
    back = [L.og_bitfield_permut_to_index(n, k, L.og_index_to_bitfield_permut(n, k, i)) for i in idx]
    assert idx == back


def test_bitfield_permut_idx_roundtrip(kat):
    n, k = kat["roundtrip"]["bitfield_to_index_to_bitfield"]
    L = og.lib()
    bf = [L.og_index_to_bitfield_permut(n, k, i) for i in range(L.og_n_choose_k(n, k))]
    bf2 = [L.og_index_to_bitfield_permut(n, k, L.og_bitfield_permut_to_index(n, k, b)) for b in bf]
    assert bf == bf2
    assert len(set(bf)) == len(bf) and all(bin(b).count("1") == k for b in bf)


# ---- src/grade_set.rs:338-373 ------------------------------------------------------------
def _gs(grades):
    g = E()
    for k in grades:
        g = g + S(k)
    return g


def test_gs_neq():
    assert S(3) != S(4)


def test_gs_neg_grade_is_empty():
    assert S(-1) == E()


def test_gs_add_self_id():
    assert S(3) + S(3) == S(3)


def test_gs_add_empty_id():
    assert S(3) + E() == S(3)


def test_gs_mul_empty_absorb():
    assert S(3) * E() == E()


@pytest.mark.parametrize("name", ["mul_scal_id", "mul_vecs", "mul_bivec_quadvec",
                                  "mul_trivec_quadvec", "mul_trivec_pentavec"])
def test_gs_mul_single(kat, name):
    c = kat["grade_set"][name]
    assert S(c["a"]) * S(c["b"]) == _gs(c["expected"])


def test_gs_mul_vec_rotor(kat):
    c = kat["grade_set"]["mul_vec_rotor"]
    assert _gs(c["a"]) * _gs(c["b"]) == _gs(c["expected"])


def test_gs_range(kat):
    c = kat["grade_set"]["range"]
    assert R(c["x"], c["y"]) == _gs(c["expected"])


def test_gs_intersect(kat):
    c = kat["grade_set"]["intersect"]
    assert R(*c["a"]).intersection(R(*c["b"])) == R(*c["expected"])


def test_gs_single_graded():
    assert (S(1) + S(1)).is_single() is True


def test_gs_not_single_graded():
    assert (S(1) + S(2)).is_single() is False


def test_gs_empty_not_single_graded():
    assert E().is_single() is False


def test_gs_empty_intersection_is_empty():
    assert S(0).intersection(S(1)).is_empty() is True


def test_gs_iter_grades(kat):
    c = kat["grade_set"]["iter_grades"]
    g = E()
    for k in c["grades"]:
        g = g + S(k)
    assert g.iter() == c["expected"]


def test_gs_parts_contributing_to_geom_prod(kat):
    c = kat["grade_set"]["parts_contributing_to_geom_prod"]
    l, r = _gs(c["self"]).parts_contributing_to_product(og.SEL_GEOMETRIC, _gs(c["left"]), _gs(c["right"]))
    assert l == _gs(c["expected_left"]) and r == _gs(c["expected_right"])


def test_gs_parts_contributing_to_outer_prod(kat):
    c = kat["grade_set"]["parts_contributing_to_outer_prod"]
    l, r = _gs(c["self"]).parts_contributing_to_product(og.SEL_OUTER, _gs(c["left"]), _gs(c["right"]))
    assert l == _gs(c["expected_left"]) and r == _gs(c["expected_right"])


# ---- src/graded.rs:230-232 ---------------------------------------------------------------
def test_hash_map_mv_eq(kat):
    d = {int(k): v for k, v in kat["graded"]["hash_map_mv_eq"].items()}
    assert grade_map_mv(d) == grade_map_mv(d)
    assert grade_map_mv(d) != grade_map_mv({1: [1, 2, 4]})
    assert grade_map_mv(d) != grade_map_mv({2: [1, 2, 3]})


def test_f32_mode_rounds_every_operation_to_binary32():
    """OG_EVAL_F32 is not `compute in f64, round at the end`: a sum whose f64 and f32 evaluations differ must come out as the
    sequential binary32 evaluation (numpy float32 arithmetic, the same statements in the same order: eval.rs:82)."""
    rng = np.random.default_rng(5)
    a = rng.uniform(-1, 1, 8).astype(np.float32)
    b = rng.uniform(-1, 1, 8).astype(np.float32)
    full = [0, 1, 2, 3]
    mk = lambda v: og.mv(grade_map_mv({0: [float(v[0])], 1: [float(x) for x in v[1:4]], 2: [float(x) for x in v[4:7]], 3: [float(v[7])]}))
    spec = (mk(a) * mk(b)).specialize(3)
    out32 = spec.eval(mode=og.EVAL_F32)
    out64 = spec.eval()
    # sequential binary32 evaluation of the same list
    root = spec.nodes()[spec.root()]
    res = {k: np.zeros(len(out64.grade_slice(k)), dtype=np.float32) for k in full}
    sl = lambda v, k: {0: v[0:1], 1: v[1:4], 2: v[4:7], 3: v[7:8]}[k]
    la = np.float32(0) + a   # the operand copies of eval.rs:27-31 (0.0 + x)
    lb = np.float32(0) + b
    for (lg, li, rg, ri, og_, oi, coeff) in spec.comp_muls(spec.root()):
        prod = np.float32(sl(la, lg)[li] * sl(lb, rg)[ri])
        prod = np.float32(prod * np.float32(coeff))
        res[og_][oi] = np.float32(res[og_][oi] + prod)
    differs = False
    for k in full:
        got = np.asarray(out32.grade_slice(k))
        assert np.array_equal(got, res[k].astype(np.float64)), (k, got, res[k])
        assert np.array_equal(got, got.astype(np.float32).astype(np.float64))   # every value is a binary32 value
        differs |= not np.array_equal(got, np.asarray(out64.grade_slice(k)).astype(np.float32).astype(np.float64))
    assert differs, "the case does not tell the f32 mode from a rounded f64 evaluation"
    assert root.minimal == 0xF
