"""The exp / log EXTENSION of the oracle (OG_EVAL_EXT_EXPLOG): "no reference behaviour, parity unpinned" -- the reference's
eval.rs:112-113 is todo!().  What can be checked is mathematics: exp(B) exp(-B) = 1, exp(log R) = R on unit versors,
agreement with the Taylor series evaluated by the oracle's own (reference-pinned) product, sqrt(R)^2 = R, and the domain
rule (a k-vector whose square is not scalar is refused).  Without the flag the arms panic like the reference."""
import numpy as np
import pytest

from oracle import pyoracle as og

EXT = og.EVAL_EXT_EXPLOG
R3 = 3
CGA = [1.0, 1.0, 1.0, 1.0, -1.0]
STA = [1.0, -1.0, -1.0, -1.0]


def _vec(alg, v):
    return og.mv(og.GradeMapMV({1: np.asarray(v, dtype=np.float64)}))


def _simple_bivector(alg, rng, scale=1.0):
    """u ^ v as a value (grade 2 only), evaluated by the oracle itself"""
    n = alg if isinstance(alg, int) else len(alg)
    u, v = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)
    out = (_vec(alg, u) ^ _vec(alg, v)).specialize(alg).eval().to_dict()
    return {2: out[2] * scale}


def _close(a, b, tol):
    assert set(a) == set(b)
    for k in a:
        assert np.allclose(a[k], b[k], rtol=0, atol=tol), (k, a[k], b[k])


@pytest.mark.parametrize("alg", [R3, CGA, STA])
def test_exp_times_exp_of_minus_is_one(alg):
    rng = np.random.default_rng(1)
    for _ in range(5):
        B = _simple_bivector(alg, rng, 1.3)
        b = og.mv(og.GradeMapMV(B))
        nb = og.mv(og.GradeMapMV({2: -B[2]}))
        out = (b.exp() * nb.exp()).specialize(alg).eval(mode=EXT).to_dict()
        assert abs(out[0][0] - 1.0) < 1e-13
        for k in out:
            if k != 0:
                assert np.abs(out[k]).max() < 1e-13


@pytest.mark.parametrize("alg", [R3, CGA, STA])
def test_exp_matches_the_taylor_series_of_the_oracles_own_product(alg):
    rng = np.random.default_rng(2)
    B = _simple_bivector(alg, rng, 0.9)
    got = og.mv(og.GradeMapMV(B)).exp().specialize(alg).eval(mode=EXT).to_dict()
    # sum_{n < 24} B^n / n!, powers by repeated products evaluated WITHOUT the extension
    n = alg if isinstance(alg, int) else len(alg)
    term = {0: np.array([1.0])}
    total = {0: np.array([1.0]), 2: np.zeros_like(B[2])}
    for i in range(1, 24):
        prod = (og.mv(og.GradeMapMV(term)) * og.mv(og.GradeMapMV(B))).specialize(alg).eval().to_dict()
        term = {k: v / i for k, v in prod.items() if k in (0, 2)}       # B is simple: higher grades vanish
        assert all(np.abs(v).max() < 1e-12 for k, v in prod.items() if k not in (0, 2))
        for k, v in term.items():
            total[k] = total[k] + v
    _close(got, total, 1e-13)


@pytest.mark.parametrize("alg,scale", [(R3, 1.0), (CGA, 0.7), (STA, 0.8)])
def test_exp_of_log_is_the_identity_on_unit_versors(alg, scale):
    rng = np.random.default_rng(3)
    B = _simple_bivector(alg, rng, scale)
    R = og.mv(og.GradeMapMV(B)).exp().specialize(alg).eval(mode=EXT).to_dict()      # unit versor, grades {0, 2}
    back = og.mv(og.GradeMapMV(R)).log().exp().specialize(alg).eval(mode=EXT).to_dict()
    _close(back, R, 1e-13)
    # ... and log(exp(B)) = B while the angle stays below pi
    logged = og.mv(og.GradeMapMV(R)).log().specialize(alg).eval(mode=EXT).to_dict()
    _close(logged, B, 1e-13)


def test_sqrt_of_a_rotor_squares_back_to_it():
    """expr.rs:300-319: sqrt of a non-scalar = pow(0.5) = exp(log(R) * 0.5)"""
    rng = np.random.default_rng(4)
    B = _simple_bivector(R3, rng, 1.1)
    R = og.mv(og.GradeMapMV(B)).exp().specialize(R3).eval(mode=EXT).to_dict()
    h = og.mv(og.GradeMapMV(R)).sqrt()
    sq = (h * h).specialize(R3).eval(mode=EXT).to_dict()
    _close({k: v for k, v in sq.items() if np.abs(v).max() > 1e-13}, R, 1e-13)


def test_null_and_hyperbolic_generators():
    # e1 ^ (e4 + e5) squares to 0 in R^{4,1}: exp = 1 + B
    B = {2: np.zeros(10)}
    out = (_vec(CGA, [1, 0, 0, 0, 0]) ^ _vec(CGA, [0, 0, 0, 1, 1])).specialize(CGA).eval().to_dict()
    got = og.mv(og.GradeMapMV({2: out[2]})).exp().specialize(CGA).eval(mode=EXT).to_dict()
    assert got[0][0] == 1.0 and np.array_equal(got[2], out[2])
    # e1 e5 squares to +1: cosh / sinh
    b = (_vec(CGA, [1, 0, 0, 0, 0]) ^ _vec(CGA, [0, 0, 0, 0, 1])).specialize(CGA).eval().to_dict()[2] * 0.5
    got = og.mv(og.GradeMapMV({2: b})).exp().specialize(CGA).eval(mode=EXT).to_dict()
    assert got[0][0] == np.cosh(0.5) and np.allclose(got[2], b * (np.sinh(0.5) / 0.5), rtol=0, atol=1e-16)
    del B


def test_exp_of_a_bare_scalar_is_a_reference_panic_at_specialization():
    """grade_set.rs:181 lets `exp` through on a grade-0 operand (maximal set {0}), but specialization hands the child
    `wanted.log()` (specialize.rs:91) and {0}.log() asserts (grade_set.rs:190-197): upstream panics, so do the oracle and
    the host mirror.  Only a hand-built flat program can reach the evaluation of exp(scalar): there the extension
    computes e^a (tests/test_gpu_explog.py::test_exp_of_a_bare_scalar_through_the_raw_abi)."""
    with pytest.raises(og.OraclePanic) as ei:
        og.mv(og.GradeMapMV({0: [0.75]})).exp().specialize(3)
    assert "log can only be used" in str(ei.value)
    import gaast_amd as ga
    with pytest.raises(ga.GaastError) as ei:
        ga.mv(ga.Input(0, [0], 3)).exp().specialize(3)
    assert "log can only be used" in str(ei.value)


def test_a_bivector_whose_square_is_not_scalar_is_refused():
    B = {2: np.array([1.0, 0.0, 0.0, 0.0, 0.0, 1.0])}        # e12 + e34 in R^4: B B has an e1234 part
    with pytest.raises(og.OraclePanic) as ei:
        og.mv(og.GradeMapMV(B)).exp().specialize(4).eval(mode=EXT)
    assert ei.value.code == og.PANIC_DOMAIN


def test_without_the_flag_the_arms_panic_like_the_reference():
    with pytest.raises(og.OraclePanic) as ei:
        og.mv(og.GradeMapMV({2: [0.1, 0.2, 0.3]})).exp().specialize(3).eval()
    assert ei.value.code == 2
    with pytest.raises(og.OraclePanic) as ei:
        og.mv(og.GradeMapMV({0: [1.0], 2: [0.1, 0.2, 0.3]})).log().specialize(3).eval(mode=og.EVAL_DEBUG)
    assert ei.value.code == 2
