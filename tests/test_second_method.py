"""The independent bitmask XOR-convolution agrees with the oracle (CPU), and the textbook
identities of SURVEY.md 8c hold for the oracle: e_i e_i = g_ii, anticommutation, R^3 cross
product, Cl(0,1) = C, Cl(0,2) = H, associativity, rotor normalisation, determinant."""
import numpy as np
import pytest

from helpers import (OracleBackend, bits_to_row, full_grades, gp_bits, n_choose_k, random_mv, row_to_bits)
from oracle import pyoracle as og


def _oracle_gp(n, metric, a, b, grades_a, grades_b):
    B = OracleBackend()
    spec = (B.value(a) * B.value(b)).specialize(metric)
    return spec.eval().to_dict()


@pytest.mark.parametrize("metric", [[1.0] * 5, [1.0, 1.0, 1.0, 1.0, -1.0], [0.0, 1.0, 1.0], [2.0, -0.5, 3.0, 0.25],
                                    [1.0] * 6, [1.0, -1.0, 1.0, 0.0, -1.0, 1.0, 1.0], [1.0] * 8,
                                    [-1.0, 1.0, 1.0, 0.0, 1.0, -1.0, 1.0, 1.0], [-1.0] * 8])
def test_bitmask_convolution_matches_oracle(metric):
    n = len(metric)
    rng = np.random.default_rng(21)
    a, b = random_mv(rng, n, full_grades(n)), random_mv(rng, n, full_grades(n))
    out = _oracle_gp(n, metric, a, b, None, None)
    ra = np.concatenate([a[k] for k in sorted(a)])
    rb = np.concatenate([b[k] for k in sorted(b)])
    C = gp_bits(n, metric, row_to_bits(n, full_grades(n), ra), row_to_bits(n, full_grades(n), rb))
    S = gp_bits(n, metric, row_to_bits(n, full_grades(n), ra), row_to_bits(n, full_grades(n), rb), absolute=True)
    got = np.concatenate([out[k] for k in sorted(out)])
    want = bits_to_row(n, sorted(out), C)
    bound = 4 * 2.0 ** -53 * bits_to_row(n, sorted(out), S) + 1e-300
    assert np.all(np.abs(got - want) <= bound)


def test_basis_vector_squares_and_anticommutation():
    metric = [1.0, -1.0, 0.0, 2.5]
    es = og.Expr.basis_vectors(4)
    for i in range(4):
        out = (es[i] * es[i]).specialize(metric).eval().to_dict()
        assert out[0][0] == metric[i]
        for j in range(i + 1, 4):
            ij = (es[i] * es[j]).specialize(metric).eval().to_dict()
            ji = (es[j] * es[i]).specialize(metric).eval().to_dict()
            assert np.array_equal(ij[2], -ji[2]) and ij[0][0] == 0.0


def test_r3_vector_product_is_dot_plus_cross_dual():
    rng = np.random.default_rng(2)
    u, v = rng.uniform(-1, 1, 3), rng.uniform(-1, 1, 3)
    out = (og.mv(og.GradeMapMV({1: u})) * og.mv(og.GradeMapMV({1: v}))).specialize(3).eval().to_dict()
    assert np.isclose(out[0][0], u @ v, rtol=0, atol=1e-15)
    cx = np.cross(u, v)  # grade 2 is [e12, e13, e23] = [cz, -cy, cx]
    assert np.allclose(out[2], [cx[2], -cx[1], cx[0]], rtol=0, atol=1e-15)


def test_cl01_is_complex_and_cl02_is_quaternions():
    z1 = og.mv(og.GradeMapMV({0: [2.0], 1: [3.0]}))
    z2 = og.mv(og.GradeMapMV({0: [-1.0], 1: [0.5]}))
    out = (z1 * z2).specialize([-1.0]).eval().to_dict()
    w = complex(2, 3) * complex(-1, 0.5)
    assert out[0][0] == w.real and out[1][0] == w.imag
    # i = e1, j = e2, k = e1 e2 in Cl(0,2): i^2 = j^2 = k^2 = ijk = -1
    e1, e2 = og.Expr.basis_vectors(2)
    k = e1 * e2
    for q in (e1 * e1, e2 * e2, k * k, e1 * e2 * k):
        assert q.specialize([-1.0, -1.0]).eval().to_dict()[0][0] == -1.0


def test_associativity_and_rotor_norm():
    rng = np.random.default_rng(4)
    n = 4
    a, b, c = (og.mv(og.GradeMapMV(random_mv(rng, n, full_grades(n)))) for _ in range(3))
    l = ((a * b) * c).specialize(n).eval().to_dict()
    r = (a * (b * c)).specialize(n).eval().to_dict()
    for k in l:
        assert np.allclose(l[k], r[k], rtol=0, atol=1e-13)
    vs = []
    for _ in range(4):
        v = rng.uniform(-1, 1, n)
        vs.append(og.mv(og.GradeMapMV({1: v / np.linalg.norm(v)})))
    R = vs[0] * vs[1] * vs[2] * vs[3]
    out = (R * R.rev()).specialize(n).eval().to_dict()
    assert abs(out[0][0] - 1.0) < 1e-14
    assert all(np.all(np.abs(out[k]) < 1e-14) for k in out if k != 0)


def test_outer_product_of_n_vectors_is_determinant():
    rng = np.random.default_rng(6)
    n = 4
    M = rng.uniform(-1, 1, (n, n))
    e = og.mv(og.GradeMapMV({1: M[0]}))
    for i in range(1, n):
        e = e ^ og.mv(og.GradeMapMV({1: M[i]}))
    out = e.specialize(n).eval().to_dict()
    assert np.isclose(out[n][0], np.linalg.det(M), rtol=1e-12, atol=1e-14)
