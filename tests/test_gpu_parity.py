"""Parity of the HIP evaluator (through the C ABI) with the oracle, on a real MI355X.

Bar (BASELINE.json north_star): bit-exact for grade/index bookkeeping; f64 component values
bit-exact on the exact kernels (same order of roundings as src/eval.rs:82), and within
    |err| <= 4 * eps(dtype) * sum|terms|      (per output component)
on the re-ordered dense kernel and for the f32 extension, with eps = machine epsilon of the
dtype (2^-52 for f64, 2^-23 for f32) and sum|terms| the sum of |left * right * coeff| over the
comp-mul entries of that component.  (Measured worst case over the suite: 1.9 eps sum|terms|.)
"""
import numpy as np
import pytest

import gaast_amd as ga
from exprs import CASES, CGA
from helpers import (HipBackend, OracleBackend, bits_to_row, full_grades, gp_bits, hip_eval_batch,
                     n_choose_k, oracle_eval_batch, random_mv, row_to_bits, rows_of, split_row)
from oracle import pyoracle as og

pytestmark = pytest.mark.gpu


def _oracle_value(name, seed=7):
    alg, build = CASES[name]
    return build(OracleBackend(), np.random.default_rng(seed)).specialize(alg).eval()


def _hip_value(name, seed=7, **kw):
    alg, build = CASES[name]
    return build(HipBackend(), np.random.default_rng(seed)).specialize(alg, **kw).eval()


def _assert_map_equal(h, o):
    hd, od = h.to_dict(), o.to_dict()
    assert set(hd) == set(od), f"grades {sorted(hd)} vs {sorted(od)}"
    for k in od:
        assert np.array_equal(hd[k], od[k]), f"grade {k}: {hd[k]} vs {od[k]}"
        assert np.array_equal(np.signbit(hd[k]), np.signbit(od[k])), f"grade {k}: sign of zero"


# ---- every catalogue expression, one item, f64, exact kernels: bit-exact ----------------------
@pytest.mark.parametrize("name", sorted(CASES))
def test_exact_path_bit_exact(name):
    """default plan for small programs: hiprtc-specialised straight-line kernel (ast_jit)"""
    _assert_map_equal(_hip_value(name, flags=ga.FLAG_EXACT_ORDER), _oracle_value(name))


@pytest.mark.parametrize("name", sorted(CASES))
def test_interpreter_kernel_bit_exact(name):
    """same plans on the LDS interpreter kernel (ast_fused), no run-time compilation"""
    _assert_map_equal(_hip_value(name, flags=ga.FLAG_EXACT_ORDER | ga.FLAG_NO_JIT), _oracle_value(name))


@pytest.mark.parametrize("name", sorted(CASES))
def test_unfused_plan_bit_exact(name):
    """One launch per eval.rs arm, every operand materialised exactly like the reference."""
    _assert_map_equal(_hip_value(name, flags=ga.FLAG_NO_FUSION), _oracle_value(name))


@pytest.mark.parametrize("name", ["kat_vecs_to_bivec", "kat_vecs_to_trivec", "kat_vec_norm", "kat_projection"])
def test_reference_known_answers_on_gpu(name, golden_dir):
    """expr_eq! of src/eval.rs:122-163 with the GPU as the evaluator."""
    import json, os
    kat = json.load(open(os.path.join(golden_dir, "ref_kat.json")))["eval"][name[4:]]
    want = ga.grade_map_mv({int(k): v for k, v in kat["expected"].items()})
    assert _hip_value(name) == want


# ---- batched evaluation: BASELINE config 1 and 5 ----------------------------------------------
def _cfg1(B):
    a, b, c = (B.input(s, full_grades(3), 3) for s in range(3))
    return (a + b * c).g(2)


@pytest.mark.parametrize("flags,kernel", [(0, "ast_jit"), (ga.FLAG_NO_JIT, "ast_fused"), (ga.FLAG_NO_FUSION, "product_csr")])
@pytest.mark.parametrize("batch", [1, 2, 63, 257, 4096])
def test_cfg1_batched_bit_exact(batch, flags, kernel):
    rng = np.random.default_rng(1)
    rows = {s: rows_of(3, full_grades(3), batch, rng) for s in range(3)}
    want, omask = oracle_eval_batch(_cfg1, 3, rows, batch)
    got, hmask, spec = hip_eval_batch(_cfg1, 3, rows, batch, flags=flags)
    assert hmask == omask == 0b100
    assert any(kernel in l for l in spec.launches()), spec.launches()
    assert np.array_equal(got, want)


def _sandwich(B, g1=False):
    r = B.input(0, [0, 2, 4], 5)
    x = B.input(1, [1], 5)
    e = r * x * r.rev()
    return e.g(1) if g1 else e


@pytest.mark.parametrize("flags", [0, ga.FLAG_NO_JIT])
@pytest.mark.parametrize("g1", [False, True])
@pytest.mark.parametrize("shared_r", [False, True])
def test_cfg5_sandwich_batched_bit_exact(g1, shared_r, flags):
    batch = 1000
    rng = np.random.default_rng(5)
    rows = {0: rows_of(5, [0, 2, 4], 1 if shared_r else batch, rng), 1: rows_of(5, [1], batch, rng)}
    build = lambda B: _sandwich(B, g1)
    want, omask = oracle_eval_batch(build, CGA, rows, batch)
    got, hmask, spec = hip_eval_batch(build, CGA, rows, batch, flags=flags)
    assert hmask == omask
    if shared_r and flags == 0:
        # round 4: the rotor is shared by all items, the program is arithmetic-bound (336 / 160 comp-muls against 168 / 80 bytes per
        # item) and nothing asks for the reference's bits: the specialised kernel's contracted variant runs (l * r + acc as one fused
        # multiply-add: one rounding per term instead of two).  Within 4 eps sum |terms| -- bounded here by (sum |R_i|)^2 sum |X_j| --
        # and, with GAAST_FLAG_EXACT_ORDER, the bits again.
        assert any("fused multiply-adds under shared operands" in l for l in spec.launches()), spec.launches()
        bound = 4 * np.finfo(np.float64).eps * np.abs(rows[0]).sum() ** 2 * np.abs(rows[1]).sum(axis=1)
        assert np.all(np.abs(got - want) <= bound[:, None]), float(np.abs(got - want).max())
        assert not np.array_equal(got, want)   # (the variant is really the one that ran)
        got, hmask, spec = hip_eval_batch(build, CGA, rows, batch, flags=ga.FLAG_EXACT_ORDER)
        assert not any("fused multiply-adds" in l for l in spec.launches()), spec.launches()
    assert np.array_equal(got, want)
    # Q4: the debug-build behaviour of the reference is a panic, reported as a status
    with pytest.raises(ga.GaastError) as ei:
        hip_eval_batch(build, CGA, rows, batch, flags=ga.FLAG_DEBUG_OVERFLOW)
    assert ei.value.status_name == "OVERFLOW"
    with pytest.raises(og.OraclePanic):
        oracle_eval_batch(build, CGA, rows, 1, mode=og.EVAL_DEBUG)


def test_true_rotor_sandwich_preserves_norm():
    """R = product of 4 unit vectors of R^{4,1}: R X ~R keeps X.X (f64, 1e-12)."""
    rng = np.random.default_rng(9)
    metric = np.array(CGA)
    batch = 512
    vs = rng.uniform(-1, 1, (4, 5))
    vs /= np.sqrt(np.abs((vs * vs) @ metric))[:, None]
    B = OracleBackend()
    R = None
    for v in vs:
        e = B.value({1: v})
        R = e if R is None else R * e
    rmap = R.specialize(CGA).eval().to_dict()
    rrow = np.concatenate([rmap[k] for k in (0, 2, 4)])[None, :]
    rows = {0: rrow, 1: rows_of(5, [1], batch, rng)}
    got, _, _ = hip_eval_batch(lambda B: _sandwich(B, True), CGA, rows, batch)
    n_in = (rows[1] ** 2) @ metric
    n_out = (got ** 2) @ metric
    scale = ((rmap[0] ** 2).sum() + (rmap[2] ** 2).sum() + (rmap[4] ** 2).sum())
    assert np.allclose(n_out, n_in * 1.0, rtol=1e-9 * max(1.0, scale), atol=1e-9)


# ---- dense kernel: tolerance against the oracle and the independent bitmask method --------------
def _gp(n):
    def build(B):
        return B.input(0, full_grades(n), n) * B.input(1, full_grades(n), n)
    return build


def _dense_bound(n, metric, ra, rb, eps):
    S = gp_bits(n, metric, np.abs(row_to_bits(n, full_grades(n), ra)), np.abs(row_to_bits(n, full_grades(n), rb)),
                absolute=True)
    return 4 * eps * bits_to_row(n, full_grades(n), S) + 1e-300


@pytest.mark.parametrize("n,dtype", [(6, ga.F64), (6, ga.F32), (7, ga.F64), (8, ga.F32), (8, ga.F64), (9, ga.F32), (10, ga.F32)])
def test_dense_gp_matches_oracle(n, dtype):
    batch = 37 if n <= 6 else 5
    rng = np.random.default_rng(2)
    npdt = np.float32 if dtype == ga.F32 else np.float64
    rows = {0: rows_of(n, full_grades(n), batch, rng, npdt), 1: rows_of(n, full_grades(n), batch, rng, npdt)}
    want, _ = oracle_eval_batch(_gp(n), n, rows, batch)
    got, mask, spec = hip_eval_batch(_gp(n), n, rows, batch, dtype=dtype)
    assert any("product_dense" in l for l in spec.launches()), spec.launches()
    eps = 2.0 ** -23 if dtype == ga.F32 else 2.0 ** -52
    for i in range(batch):
        bound = _dense_bound(n, [1.0] * n, rows[0][i], rows[1][i], eps)
        assert np.all(np.abs(got[i].astype(np.float64) - want[i]) <= bound), f"item {i}"


@pytest.mark.parametrize("metric", [[1.0, 1.0, 1.0, 1.0, -1.0, 1.0, -1.0], [1.0, 1.0, 1.0, 1.0, 0.0, 1.0, -1.0, 0.0]])
def test_dense_gp_mixed_signature(metric):
    n = len(metric)
    batch = 9
    rng = np.random.default_rng(3)
    rows = {0: rows_of(n, full_grades(n), batch, rng), 1: rows_of(n, full_grades(n), batch, rng)}
    want, _ = oracle_eval_batch(_gp(n), metric, rows, batch)
    got, mask, spec = hip_eval_batch(_gp(n), metric, rows, batch)
    assert any("product_dense" in l for l in spec.launches())
    for i in range(batch):
        bound = _dense_bound(n, np.abs(metric), rows[0][i], rows[1][i], 2.0 ** -52)
        assert np.all(np.abs(got[i] - want[i]) <= bound)


@pytest.mark.parametrize("n,metric", [(8, [1.0, 1.0, 1.0, 1.0, -1.0, 1.0, -1.0, 1.0]), (8, [1.0, 1.0, 1.0, 1.0, 0.0, 1.0, -1.0, 0.0]),
                                      (9, [1.0, 1.0, 1.0, 1.0, -1.0, -1.0, 1.0, 0.0, 1.0])])
def test_dense_mfma16_mixed_and_degenerate_signatures_f32(n, metric):
    """k_gp_mfma16 (four items per wave): negative and null basis vectors among the hi bits, a batch that does not
    fill the last workgroup, and the vector-FMA kernel (GAAST_FLAG_NO_MFMA) as a second opinion"""
    batch = 37
    rng = np.random.default_rng(70 + n)
    rows = {0: rows_of(n, full_grades(n), batch, rng, np.float32), 1: rows_of(n, full_grades(n), batch, rng, np.float32)}
    want, _ = oracle_eval_batch(_gp(n), metric, rows, batch)
    got, _, spec = hip_eval_batch(_gp(n), metric, rows, batch, dtype=ga.F32)
    assert any("product_dense_mfma" in l for l in spec.launches()), spec.launches()
    alt, _, spec2 = hip_eval_batch(_gp(n), metric, rows, batch, dtype=ga.F32, flags=ga.FLAG_NO_MFMA)
    assert any(l.startswith("product_dense[") for l in spec2.launches()), spec2.launches()
    for i in range(batch):
        bound = _dense_bound(n, np.abs(metric), rows[0][i], rows[1][i], 2.0 ** -23)
        assert np.all(np.abs(got[i].astype(np.float64) - want[i]) <= bound), i
        assert np.all(np.abs(alt[i].astype(np.float64) - want[i]) <= bound), i


@pytest.mark.parametrize("n,metric", [(8, [1.0] * 8), (8, [1.0, 1.0, 1.0, 1.0, -1.0, 1.0, -1.0, 1.0]), (8, [0.0, 1.0, 1.0, 1.0, 0.0, 1.0, -1.0, 1.0]),
                                      (8, [-1.0] * 8), (9, [1.0] * 9), (9, [1.0, 1.0, -1.0, 1.0, -1.0, -1.0, 1.0, 0.0, 1.0])])
def test_dense_mfma16d_f64_against_the_oracle(n, metric):
    """k_gp_mfma16d (f64, v_mfma_f64_16x16x4_f64, one item per workgroup; n = 9: two waves per item): Euclidean, negative
    and null basis vectors among lo and hi bits, a batch larger than the resident workgroups' first pass is not needed --
    the persistent loop is exercised by 37 items on however many workgroups fit -- and the vector-FMA kernel
    (GAAST_FLAG_NO_MFMA) as a second opinion where it applies"""
    batch = 37
    rng = np.random.default_rng(170 + n)
    rows = {0: rows_of(n, full_grades(n), batch, rng), 1: rows_of(n, full_grades(n), batch, rng)}
    want, _ = oracle_eval_batch(_gp(n), metric, rows, batch)
    got, _, spec = hip_eval_batch(_gp(n), metric, rows, batch, dtype=ga.F64)
    assert any(l.startswith("product_dense_mfma[") for l in spec.launches()), spec.launches()
    for i in range(batch):
        bound = _dense_bound(n, np.abs(metric), rows[0][i], rows[1][i], 2.0 ** -52)
        assert np.all(np.abs(got[i] - want[i]) <= bound), i
    if sum(m == 1.0 for m in metric) >= 4 or sum(m == -1.0 for m in metric) >= 4:
        alt, _, spec2 = hip_eval_batch(_gp(n), metric, rows, batch, dtype=ga.F64, flags=ga.FLAG_NO_MFMA)
        assert any(l.startswith("product_dense[") for l in spec2.launches()), spec2.launches()
        for i in range(batch):
            bound = _dense_bound(n, np.abs(metric), rows[0][i], rows[1][i], 2.0 ** -52)
            assert np.all(np.abs(alt[i] - want[i]) <= bound), i


def test_dense_mfma16d_f64_basis_blades_and_partial_grades():
    """one-hot operands are reproduced exactly (every product of basis blades, sign included), and operands that hold only
    some grades / a projected result go through the general staging and the result map"""
    n = 8
    from helpers import blades_in_row_order
    blades = blades_in_row_order(n, full_grades(n))
    N = 1 << n
    rng = np.random.default_rng(8)
    batch = 64
    ia, ib = rng.integers(0, N, batch), rng.integers(0, N, batch)
    A = np.zeros((batch, N)); B = np.zeros((batch, N))
    A[np.arange(batch), ia] = 1.0
    B[np.arange(batch), ib] = -2.0
    rows = {0: A, 1: B}
    metric = [1.0, -1.0, 1.0, 1.0, -1.0, 1.0, 1.0, -1.0]
    want, _ = oracle_eval_batch(_gp(n), metric, rows, batch)
    got, _, spec = hip_eval_batch(_gp(n), metric, rows, batch, dtype=ga.F64)
    assert any(l.startswith("product_dense_mfma[") for l in spec.launches()), spec.launches()
    assert np.array_equal(got, want)
    # partial grades: even x (odd + grade 8) -> grades 1, 3 projected
    lg, rg = [0, 2, 4, 6, 8], [1, 3, 5, 7, 8]
    build = lambda Bk: (Bk.input(0, lg, n) * Bk.input(1, rg, n)).gselect([1, 3, 5])
    rows = {0: rows_of(n, lg, 5, rng), 1: rows_of(n, rg, 5, rng)}
    want, wmask = oracle_eval_batch(build, n, rows, 5)
    got, mask, spec = hip_eval_batch(build, n, rows, 5, dtype=ga.F64)
    assert mask == wmask
    if any("product_dense" in l for l in spec.launches()):
        assert np.allclose(got, want, rtol=0, atol=1e-12)
    else:
        assert np.array_equal(got, want)


def test_dense_gp_partial_grades_and_projection():
    """even * full -> grades {1,3,5}: absent operand grades are zeros, unwanted outputs dropped."""
    n = 7
    batch = 11
    rng = np.random.default_rng(8)
    even = [0, 2, 4, 6]
    build = lambda B: (B.input(0, even, n) * B.input(1, full_grades(n), n)).gselect([1, 3, 5])
    rows = {0: rows_of(n, even, batch, rng), 1: rows_of(n, full_grades(n), batch, rng)}
    want, omask = oracle_eval_batch(build, n, rows, batch)
    got, hmask, spec = hip_eval_batch(build, n, rows, batch)
    assert hmask == omask and any("product_dense" in l for l in spec.launches())
    assert np.allclose(got, want, rtol=0, atol=1e-12)


def test_dense_n12_against_bitmask_convolution():
    """BASELINE config 3 shape (R^12, f32), a few items, checked by the independent method."""
    n, batch = 12, 3
    rng = np.random.default_rng(3)
    rows = {0: rows_of(n, full_grades(n), batch, rng, np.float32), 1: rows_of(n, full_grades(n), batch, rng, np.float32)}
    got, mask, spec = hip_eval_batch(_gp(n), n, rows, batch, dtype=ga.F32)
    assert any("product_dense" in l for l in spec.launches())
    for i in range(batch):
        A, Bb = row_to_bits(n, full_grades(n), rows[0][i]), row_to_bits(n, full_grades(n), rows[1][i])
        want = bits_to_row(n, full_grades(n), gp_bits(n, [1.0] * n, A, Bb))
        bound = _dense_bound(n, [1.0] * n, rows[0][i], rows[1][i], 2.0 ** -23)
        assert np.all(np.abs(got[i].astype(np.float64) - want) <= bound)


def test_dense_n12_basis_blades_exact():
    """e_a * e_b = +-e_{a^b} exactly (no rounding involved): every sign/index of a sample of the
    4096 x 4096 Cayley table, through the full-size kernel."""
    n = 12
    rng = np.random.default_rng(12)
    batch = 256
    a_idx, b_idx = rng.integers(0, 4096, batch), rng.integers(0, 4096, batch)
    ra, rb = np.zeros((batch, 4096), np.float32), np.zeros((batch, 4096), np.float32)
    ra[np.arange(batch), a_idx] = 1.0
    rb[np.arange(batch), b_idx] = 1.0
    got, _, _ = hip_eval_batch(_gp(n), n, {0: ra, 1: rb}, batch, dtype=ga.F32)
    from helpers import blades_in_row_order
    blades = blades_in_row_order(n, full_grades(n))
    pos_of = np.zeros(4096, dtype=np.int64)
    pos_of[blades] = np.arange(4096)
    L = og.lib()
    for i in range(batch):
        a, b = int(blades[a_idx[i]]), int(blades[b_idx[i]])
        want = np.zeros(4096, np.float32)
        want[pos_of[a ^ b]] = L.og_canonical_reordering_sign(a, b)
        assert np.array_equal(got[i], want)


# ---- exact dense products on the transposed-list kernel (k_product_ell) ---------------------------
@pytest.mark.parametrize("n,neg,batch", [(7, (), 21), (8, (), 19), (8, (1, 4, 6), 8), (9, (0,), 5)])
def test_exact_order_dense_product_is_bit_exact(n, neg, batch):
    """GAAST_FLAG_EXACT_ORDER, programs too big to fuse: rows of one length with +-1 coefficients run on
    k_product_ell -- same terms, same order, same roundings as eval.rs:77-83, several items per pass."""
    metric = [-1.0 if i in neg else 1.0 for i in range(n)]
    rng = np.random.default_rng(90 + n)
    rows = {0: rows_of(n, full_grades(n), batch, rng), 1: rows_of(n, full_grades(n), batch, rng)}
    want, omask = oracle_eval_batch(_gp(n), metric, rows, batch)
    got, hmask, spec = hip_eval_batch(_gp(n), metric, rows, batch, flags=ga.FLAG_EXACT_ORDER)
    assert any("product_ell" in l for l in spec.launches()), spec.launches()
    assert hmask == omask and np.array_equal(got, want)


@pytest.mark.parametrize("n,left_grades", [(7, [0, 1, 2]), (8, [0, 1, 2, 3])])
def test_exact_order_uniform_rows_of_odd_width(n, left_grades):
    """rows of 29 / 93 entries: the 32-word chunks of k_product_ell plus its remainder loop"""
    batch = 13
    rng = np.random.default_rng(95 + n)
    build = lambda B: B.input(0, left_grades, n) * B.input(1, full_grades(n), n)
    rows = {0: rows_of(n, left_grades, batch, rng), 1: rows_of(n, full_grades(n), batch, rng)}
    want, omask = oracle_eval_batch(build, n, rows, batch)
    got, hmask, spec = hip_eval_batch(build, n, rows, batch, flags=ga.FLAG_EXACT_ORDER)
    assert any("product_ell" in l for l in spec.launches()), spec.launches()
    assert hmask == omask and np.array_equal(got, want)


def test_exact_order_sum_of_dense_products_accumulates_bit_exact():
    """a*b + c*d: the second product adds into the buffer the first one wrote (beta = 1), reference order"""
    n, batch = 8, 11
    fg = full_grades(n)
    rng = np.random.default_rng(91)
    build = lambda B: B.input(0, fg, n) * B.input(1, fg, n) + B.input(2, fg, n) * B.input(3, fg, n)
    rows = {s: rows_of(n, fg, batch, rng) for s in range(4)}
    want, _ = oracle_eval_batch(build, n, rows, batch)
    got, _, spec = hip_eval_batch(build, n, rows, batch, flags=ga.FLAG_EXACT_ORDER)
    assert sum("product_ell" in l for l in spec.launches()) == 2, spec.launches()
    assert np.array_equal(got, want)


@pytest.mark.parametrize("n,dtype", [(8, ga.F32), (8, ga.F64), (9, ga.F64), (10, ga.F32), (10, ga.F64)])
def test_sum_of_dense_products_accumulates_on_the_matrix_core_kernels(n, dtype):
    """a*b + c*d on the default (re-ordered) path: the second product adds into the buffer the first one wrote (beta = 1) --
    the read-modify-write result path of k_gp_mfma16x4<float / double> and k_gp_mfma32p; one shared operand (batch-1 row)"""
    batch = 7
    fg = full_grades(n)
    rng = np.random.default_rng(190 + n)
    npdt = np.float32 if dtype == ga.F32 else np.float64
    build = lambda B: B.input(0, fg, n) * B.input(1, fg, n) + B.input(2, fg, n) * B.input(3, fg, n)
    rows = {s: rows_of(n, fg, batch if s != 2 else 1, rng, npdt) for s in range(4)}
    want, _ = oracle_eval_batch(build, n, rows, batch)
    got, _, spec = hip_eval_batch(build, n, rows, batch, dtype=dtype)
    assert sum(l.startswith("product_dense_mfma[") for l in spec.launches()) == 2, spec.launches()
    eps = 2.0 ** -23 if dtype == ga.F32 else 2.0 ** -52
    for i in range(batch):
        bound = (_dense_bound(n, [1.0] * n, rows[0][i], rows[1][i], eps) +
                 _dense_bound(n, [1.0] * n, rows[2][0], rows[3][i], eps)) * 1.5
        assert np.all(np.abs(got[i].astype(np.float64) - want[i]) <= bound), i


def test_exact_order_degenerate_metric_keeps_the_coefficient_list():
    """a zero in the metric makes 0.0 coefficients (entries the reference still executes): k_product_csr"""
    n, batch = 7, 6
    metric = [1.0, 1.0, 0.0, 1.0, -1.0, 1.0, 1.0]
    rng = np.random.default_rng(92)
    rows = {0: rows_of(n, full_grades(n), batch, rng), 1: rows_of(n, full_grades(n), batch, rng)}
    want, _ = oracle_eval_batch(_gp(n), metric, rows, batch)
    got, _, spec = hip_eval_batch(_gp(n), metric, rows, batch, flags=ga.FLAG_EXACT_ORDER)
    assert any("product_csr" in l for l in spec.launches()), spec.launches()
    assert np.array_equal(got, want)


# ---- opt-in matrix-representation product (GAAST_FLAG_SPINOR_GEMM) ------------------------------
# Not the reference's summation: equal in exact arithmetic, so the check is the norm-wise bound the
# header states, |err_S| <= 64 eps |A|_2 |B|_2, against the float64 bitmask convolution.
# the second one and the next five cover every case of the one-plane kernel's index basis (spinor_basis.hpp):
# lambda on bit 5 with / without alpha, lambda on bit 4, lambda = 0 with / without alpha
SPINOR_METRICS = [[1.0] * 12, [-1.0, 1.0, 1.0, -1.0, -1.0, 1.0, 1.0, 1.0, -1.0, 1.0, -1.0, -1.0],
                  [-1.0, 1.0, -1.0, 1.0, 1.0, -1.0, 1.0, -1.0, -1.0, 1.0, -1.0, -1.0],      # lam=5, alpha
                  [1.0, 1.0, -1.0, -1.0, 1.0, 1.0, -1.0, -1.0, 1.0, 1.0, -1.0, -1.0],        # lam=5, no alpha
                  [1.0, 1.0, -1.0, 1.0, -1.0, -1.0, -1.0, -1.0, 1.0, 1.0, -1.0, 1.0],        # lam=4
                  [-1.0, 1.0, -1.0, 1.0, 1.0, -1.0, -1.0, 1.0, -1.0, 1.0, 1.0, -1.0],        # lam=-1, alpha
                  [1.0, -1.0] * 6]                                                            # lam=-1, no alpha
SPINOR_VARIANTS = {2: "lam=5", 3: "lam=5", 4: "lam=4", 5: "lam=-1", 6: "lam=-1"}


@pytest.mark.parametrize("metric", SPINOR_METRICS)
def test_spinor_gemm_n12_against_bitmask_convolution(metric):
    n, batch = 12, 4
    rng = np.random.default_rng(21)
    rows = {0: rows_of(n, full_grades(n), batch, rng, np.float32), 1: rows_of(n, full_grades(n), batch, rng, np.float32)}
    got, mask, spec = hip_eval_batch(_gp(n), metric, rows, batch, dtype=ga.F32, flags=ga.FLAG_SPINOR_GEMM)
    assert any("product_spinor_gemm" in l for l in spec.launches()), spec.launches()
    idx = SPINOR_METRICS.index(metric)
    if idx in SPINOR_VARIANTS:
        assert any(SPINOR_VARIANTS[idx] in l for l in spec.launches()), spec.launches()
    for i in range(batch):
        A, Bb = row_to_bits(n, full_grades(n), rows[0][i]), row_to_bits(n, full_grades(n), rows[1][i])
        want = bits_to_row(n, full_grades(n), gp_bits(n, metric, A, Bb))
        bound = 64 * 2.0 ** -23 * np.linalg.norm(rows[0][i].astype(np.float64)) * np.linalg.norm(rows[1][i].astype(np.float64))
        err = np.abs(got[i].astype(np.float64) - want).max()
        assert err <= bound, (err, bound)


@pytest.mark.parametrize("metric", SPINOR_METRICS)
def test_spinor_gemm_basis_blades_exact(metric):
    """One-hot operands: every intermediate is a small integer, so the signs, the phases i^k and the
    blade <-> Pauli-string tables are checked exactly, for all 4096 x (a sample of) 4096 pairs."""
    n = 12
    rng = np.random.default_rng(22)
    batch = 512
    a_idx, b_idx = rng.integers(0, 4096, batch), rng.integers(0, 4096, batch)
    a_idx[:64] = np.arange(64) * 64 + 1
    ra, rb = np.zeros((batch, 4096), np.float32), np.zeros((batch, 4096), np.float32)
    ra[np.arange(batch), a_idx] = 1.0
    rb[np.arange(batch), b_idx] = 1.0
    got, _, spec = hip_eval_batch(_gp(n), metric, {0: ra, 1: rb}, batch, dtype=ga.F32, flags=ga.FLAG_SPINOR_GEMM)
    assert any("product_spinor_gemm" in l for l in spec.launches())
    from helpers import blades_in_row_order
    blades = blades_in_row_order(n, full_grades(n))
    pos_of = np.zeros(4096, dtype=np.int64)
    pos_of[blades] = np.arange(4096)
    alg = ga.MetricAlgebra(metric)
    for i in range(batch):
        a, b = int(blades[a_idx[i]]), int(blades[b_idx[i]])
        res, coeff = alg.ortho_basis_blades_gp(a, b)       # algebra.rs:73-83 (host mirror, pinned by the CPU suite)
        want = np.zeros(4096, np.float32)
        want[pos_of[res]] = coeff
        assert np.array_equal(got[i], want), (i, a, b)


@pytest.mark.parametrize("n,neg,variant", [
    (7, (), None), (8, (), None), (8, (0, 3, 6), None), (9, (1, 8), None), (10, (), None), (10, (2, 3, 9), None), (11, (0, 10), None),
    # every case of the index basis (spinor_basis.hpp) on the 16 x 16 and the 32 x 32 kernels
    (8, (2, 3, 6, 7), "lam=3"), (8, (2, 3, 4), "lam=2"), (8, (0, 1, 2, 3, 4, 6), "lam=3"), (8, (0, 3, 4, 7), "lam=-1"), (8, (1, 3, 5, 7), "lam=-1"),
    (10, (2, 3, 6, 7, 8, 9), "lam=4"), (10, (0, 1, 3, 4, 8), "lam=3"), (10, (1, 2, 5, 7, 9), "lam=-1"), (10, (2, 4, 5, 8, 9), "lam=4"),
    (10, (1, 3, 5, 7, 9), "lam=-1")])
@pytest.mark.parametrize("dtype", [ga.F32, ga.F64])
def test_spinor_gemm_smaller_dimensions(n, neg, variant, dtype):
    """n = 8 and 10 on the wave-per-item kernels (f32: k_gp_spinor_wave1, f64: k_gp_spinor_wave1d), odd n as the
    subalgebra of n + 1."""
    if dtype == ga.F64 and n == 11:
        pytest.skip("n = 11 in f64 runs on k_gp_spinor12d: test_spinor_gemm_f64")
    metric = [-1.0 if i in neg else 1.0 for i in range(n)]
    batch = 67
    rng = np.random.default_rng(30 + n)
    npdt, eps = (np.float32, 2.0 ** -23) if dtype == ga.F32 else (np.float64, 2.0 ** -52)
    rows = {0: rows_of(n, full_grades(n), batch, rng, npdt), 1: rows_of(n, full_grades(n), batch, rng, npdt)}
    got, mask, spec = hip_eval_batch(_gp(n), metric, rows, batch, dtype=dtype, flags=ga.FLAG_SPINOR_GEMM)
    assert any("product_spinor_gemm" in l for l in spec.launches()), spec.launches()
    if variant:
        assert any(variant in l for l in spec.launches()), spec.launches()
    for i in range(0, batch, 11):
        A, Bb = row_to_bits(n, full_grades(n), rows[0][i]), row_to_bits(n, full_grades(n), rows[1][i])
        want = bits_to_row(n, full_grades(n), gp_bits(n, metric, A, Bb))
        bound = 64 * eps * np.linalg.norm(rows[0][i].astype(np.float64)) * np.linalg.norm(rows[1][i].astype(np.float64))
        err = np.abs(got[i].astype(np.float64) - want).max()
        assert err <= bound, (i, err, bound)


@pytest.mark.parametrize("dtype", [ga.F32, ga.F64])
@pytest.mark.parametrize("n", [8, 9, 10])
def test_spinor_gemm_smaller_dimensions_basis_blades_exact(n, dtype):
    metric = [-1.0 if i % 3 == 1 else 1.0 for i in range(n)]
    N = 1 << n
    rng = np.random.default_rng(40 + n)
    batch = 300
    a_idx, b_idx = rng.integers(0, N, batch), rng.integers(0, N, batch)
    npdt = np.float32 if dtype == ga.F32 else np.float64
    ra, rb = np.zeros((batch, N), npdt), np.zeros((batch, N), npdt)
    ra[np.arange(batch), a_idx] = 1.0
    rb[np.arange(batch), b_idx] = 1.0
    got, _, spec = hip_eval_batch(_gp(n), metric, {0: ra, 1: rb}, batch, dtype=dtype, flags=ga.FLAG_SPINOR_GEMM)
    assert any("product_spinor_gemm" in l for l in spec.launches())
    from helpers import blades_in_row_order
    blades = blades_in_row_order(n, full_grades(n))
    pos_of = np.zeros(N, dtype=np.int64)
    pos_of[blades] = np.arange(N)
    alg = ga.MetricAlgebra(metric)
    for i in range(batch):
        res, coeff = alg.ortho_basis_blades_gp(int(blades[a_idx[i]]), int(blades[b_idx[i]]))
        want = np.zeros(N, npdt)
        want[pos_of[res]] = coeff
        assert np.array_equal(got[i], want), (i,)


@pytest.mark.parametrize("n,midx", [(12, 0), (12, 1), (12, 3), (12, 4), (12, 5), (12, 6), (11, 0),
                                    (10, 0), (10, 1), (10, 3), (10, 4), (10, 5), (10, 6), (9, 1), (8, 0), (8, 1), (8, 3), (8, 4), (8, 5), (8, 6), (7, 0)])
def test_spinor_gemm_f64(n, midx):
    """the same path in f64 (the reference's value type): bound 64 * 2^-52 * |A| |B|, every index-basis case; n = 11, 12
    on k_gp_spinor12d, n = 7..10 on the wave-per-item k_gp_spinor_wave1d (16 x 16 f64 MFMA tiles)"""
    metric = SPINOR_METRICS[midx][:n]
    batch = 5
    rng = np.random.default_rng(50 + midx)
    rows = {0: rows_of(n, full_grades(n), batch, rng), 1: rows_of(n, full_grades(n), batch, rng)}
    got, mask, spec = hip_eval_batch(_gp(n), metric, rows, batch, flags=ga.FLAG_SPINOR_GEMM)
    assert any("product_spinor_gemm" in l for l in spec.launches()), spec.launches()
    for i in range(batch):
        A, Bb = row_to_bits(n, full_grades(n), rows[0][i]), row_to_bits(n, full_grades(n), rows[1][i])
        want = bits_to_row(n, full_grades(n), gp_bits(n, metric, A, Bb))
        bound = 64 * 2.0 ** -52 * np.linalg.norm(rows[0][i]) * np.linalg.norm(rows[1][i])
        err = np.abs(got[i] - want).max()
        assert err <= bound, (i, err, bound)


def test_spinor_gemm_f64_basis_blades_exact():
    n, N = 12, 4096
    metric = SPINOR_METRICS[1]
    rng = np.random.default_rng(60)
    batch = 400
    a_idx, b_idx = rng.integers(0, N, batch), rng.integers(0, N, batch)
    ra, rb = np.zeros((batch, N)), np.zeros((batch, N))
    ra[np.arange(batch), a_idx] = 1.0
    rb[np.arange(batch), b_idx] = 1.0
    got, _, spec = hip_eval_batch(_gp(n), metric, {0: ra, 1: rb}, batch, flags=ga.FLAG_SPINOR_GEMM)
    assert any("product_spinor_gemm" in l for l in spec.launches())
    from helpers import blades_in_row_order
    blades = blades_in_row_order(n, full_grades(n))
    pos_of = np.zeros(N, dtype=np.int64)
    pos_of[blades] = np.arange(N)
    alg = ga.MetricAlgebra(metric)
    for i in range(batch):
        res, coeff = alg.ortho_basis_blades_gp(int(blades[a_idx[i]]), int(blades[b_idx[i]]))
        want = np.zeros(N)
        want[pos_of[res]] = coeff
        assert np.array_equal(got[i], want), (i,)


@pytest.mark.parametrize("n,dtype", [(12, ga.F32), (12, ga.F64), (11, ga.F64), (10, ga.F32), (8, ga.F32), (7, ga.F32),
                                     (10, ga.F64), (9, ga.F64), (8, ga.F64), (7, ga.F64)])
def test_spinor_gemm_partial_grades_unary_folding_and_shared_operand(n, dtype):
    """rotor-like even operand (shared by all items), reversed on the fly, odd result only: partial tables,
    zero-filled planes, folded unary signs, a batch-1 operand -- on every kernel of the path."""
    batch, N = 3, 1 << n
    npdt, eps = (np.float32, 2.0 ** -23) if dtype == ga.F32 else (np.float64, 2.0 ** -52)
    rng = np.random.default_rng(23 + n)
    even = [k for k in range(n + 1) if k % 2 == 0]
    odd = [k for k in range(n + 1) if k % 2 == 1]
    build = lambda B: (B.input(0, even, n).rev() * B.input(1, full_grades(n), n)).gselect(odd)
    rows = {0: rows_of(n, even, 1, rng, npdt), 1: rows_of(n, full_grades(n), batch, rng, npdt)}
    spec = build(HipBackend()).specialize(n, dtype=dtype, flags=ga.FLAG_SPINOR_GEMM)
    out = spec.eval_batch([rows[0], rows[1]], batch)
    got = out.download_rows()
    assert any("product_spinor_gemm" in l for l in spec.launches()), spec.launches()
    A = row_to_bits(n, even, rows[0][0].astype(np.float64))
    for m in range(N):
        k = bin(m).count("1")
        if (k * (k - 1) // 2) % 2:
            A[m] = -A[m]
    for i in range(batch):
        full = gp_bits(n, [1.0] * n, A, row_to_bits(n, full_grades(n), rows[1][i].astype(np.float64)))
        want = bits_to_row(n, odd, full)
        bound = 64 * eps * np.linalg.norm(A) * np.linalg.norm(rows[1][i].astype(np.float64))
        assert np.abs(got[i].astype(np.float64) - want).max() <= bound


def test_spinor_gemm_accumulates_into_a_shared_result_buffer():
    """(a*b + c*d): the second product adds into the buffer the first one wrote (beta = 1 in the kernel)."""
    n, batch = 8, 9
    rng = np.random.default_rng(77)
    fg = full_grades(n)
    build = lambda B: B.input(0, fg, n) * B.input(1, fg, n) + B.input(2, fg, n) * B.input(3, fg, n)
    rows = {s: rows_of(n, fg, batch, rng, np.float32) for s in range(4)}
    got, _, spec = hip_eval_batch(build, n, rows, batch, dtype=ga.F32, flags=ga.FLAG_SPINOR_GEMM)
    assert sum("product_spinor_gemm" in l for l in spec.launches()) == 2, spec.launches()
    for i in range(batch):
        bits = [row_to_bits(n, fg, rows[s][i].astype(np.float64)) for s in range(4)]
        want = bits_to_row(n, fg, gp_bits(n, [1.0] * n, bits[0], bits[1]) + gp_bits(n, [1.0] * n, bits[2], bits[3]))
        bound = 64 * 2.0 ** -23 * (np.linalg.norm(bits[0]) * np.linalg.norm(bits[1]) + np.linalg.norm(bits[2]) * np.linalg.norm(bits[3]))
        assert np.abs(got[i].astype(np.float64) - want).max() <= bound


@pytest.mark.parametrize("n", [8, 10, 11, 12])
@pytest.mark.parametrize("layout", ["strided_aligned", "strided_odd", "misaligned_base"])
def test_spinor_gemm_row_io_forms_give_the_same_bits(n, layout):
    """The matrix-representation kernels move rows as 16-byte pieces when base and stride allow it (k_gp_spinor12s: its FAST
    instantiation) and component by component otherwise: the arithmetic is the same, so wrapped memory with padded, odd or
    shifted rows must reproduce the contiguous result bit for bit and leave the padding alone."""
    torch = pytest.importorskip("torch")
    batch, N = 5, 1 << n
    rng = np.random.default_rng(100 + n)
    rows = {0: rows_of(n, full_grades(n), batch, rng, np.float32), 1: rows_of(n, full_grades(n), batch, rng, np.float32)}
    want, mask, spec = hip_eval_batch(_gp(n), [1.0] * n, rows, batch, dtype=ga.F32, flags=ga.FLAG_SPINOR_GEMM)
    assert any("product_spinor_gemm" in l for l in spec.launches()), spec.launches()
    pad = {"strided_aligned": 8, "strided_odd": 3, "misaligned_base": 0}[layout]   # floats of padding per row
    shift = 1 if layout == "misaligned_base" else 0                                 # base pointer off by 4 bytes
    keep, ins = [], []
    for s_ in range(2):
        flat = torch.full((batch * (N + pad) + 4,), 777.0, dtype=torch.float32, device="cuda")
        view = flat[shift:shift + batch * (N + pad)].view(batch, N + pad)
        view[:, :N] = torch.from_numpy(rows[s_]).cuda()
        keep.append(flat)
        ins.append(ga.DeviceMV.wrap_tensor(view[:, :N], n, full_grades(n)))
    oflat = torch.full((batch * (N + pad) + 4,), -5.0, dtype=torch.float32, device="cuda")
    oview = oflat[shift:shift + batch * (N + pad)].view(batch, N + pad)
    out = ga.DeviceMV.wrap_tensor(oview[:, :N], n, full_grades(n))
    spec.eval_batch(ins, batch, out=out)
    torch.cuda.synchronize()
    got = oview[:, :N].cpu().numpy()
    assert np.array_equal(got.view(np.uint32), np.asarray(want, dtype=np.float32).view(np.uint32))
    if pad:
        assert torch.all(oview[:, N:] == -5.0)
    assert float(oflat[0]) == -5.0 or shift == 0


@pytest.mark.parametrize("name", ["cfg1_r3", "cfg5_sandwich", "r6_gp_full", "weird_metric_gp"])
def test_spinor_flag_is_ignored_where_it_does_not_apply(name):
    """f64, other dimensions, degenerate metrics: the flag changes nothing."""
    _assert_map_equal(_hip_value(name, flags=ga.FLAG_SPINOR_GEMM), _hip_value(name))


def test_hiprtc_failure_falls_back_to_kernels_that_need_no_compiler():
    """GAAST_FLAG_DEBUG_JIT_FAILS makes every run-time compilation fail: small programs run on the LDS interpreter,
    plans that only fit the specialised kernel are rebuilt unfused -- same bits either way."""
    fl = ga.FLAG_EXACT_ORDER | ga.FLAG_DEBUG_JIT_FAILS
    for name in ("cfg5_sandwich", "r5_gp_full", "shared_subexpr"):
        alg, build = CASES[name]
        spec = build(HipBackend(), np.random.default_rng(7)).specialize(alg, flags=fl)
        got = spec.eval()
        assert not any("ast_jit" in l for l in spec.launches()), spec.launches()
        _assert_map_equal(got, _oracle_value(name))
    alg, build = CASES["r5_gp_full"]
    spec = build(HipBackend(), np.random.default_rng(7)).specialize(alg, flags=fl)
    assert any("product_ell" in l or "ast_fused" in l for l in spec.launches()), spec.launches()


def test_kept_jit_source_is_the_program_in_reference_order():
    """GAAST_FLAG_DEBUG_KEEP_JIT_SOURCE: the generated kernel can be read back (and is empty without the flag)."""
    alg, build = CASES["cfg5_sandwich"]
    spec = build(HipBackend(), np.random.default_rng(7)).specialize(alg, flags=ga.FLAG_DEBUG_KEEP_JIT_SOURCE)
    src = spec.jit_source()
    assert "gaast_jit" in src and "acc = acc" in src
    assert build(HipBackend(), np.random.default_rng(7)).specialize(alg).jit_source() == ""


# ---- storage, errors, edges ------------------------------------------------------------------
def test_upload_download_per_grade_roundtrip():
    rng = np.random.default_rng(0)
    m = ga.DeviceMV.alloc(5, [0, 2, 5], 33)
    assert np.array_equal(m.download_rows(), np.zeros((33, 12)))      # init_null_mv
    for k in (0, 2, 5):
        v = rng.uniform(-1, 1, (33, n_choose_k(5, k)))
        m.upload(k, v)
        assert np.array_equal(m.download(k), v)
    with pytest.raises(ga.GaastError) as ei:
        m.download(1)
    assert ei.value.status_name == "MISSING_GRADE"


def test_exp_log_report_unimplemented():
    a = ga.mv(ga.GradeMapMV({2: [0.1, 0.2, 0.3]}))
    with pytest.raises(ga.GaastError) as ei:
        a.exp().specialize(3).eval()
    assert ei.value.status_name == "UNIMPLEMENTED"
    with pytest.raises(og.OraclePanic) as eo:
        og.mv(og.GradeMapMV({2: [0.1, 0.2, 0.3]})).exp().specialize(3).eval()
    assert eo.value.code == 2


def test_q2_projection_leak_matches_reference_behaviour():
    """`p*p + p.g(0)` with shared p leaks/needs grades the projection drops; whatever
    the reference does (value or panic) the GPU path does the same."""
    for name in ("shared_subexpr",):
        try:
            want = _oracle_value(name)
        except og.OraclePanic as p:
            with pytest.raises(ga.GaastError):
                _hip_value(name)
            continue
        _assert_map_equal(_hip_value(name), want)


def test_batch_zero_and_empty_output():
    spec = _cfg1(HipBackend()).specialize(3)
    out = spec.eval_batch([np.zeros((0, 8))] * 3, 0)
    assert out.download_rows().shape == (0, 3)
    z = ga.Expr._lift(0) * ga.Expr.basis_vectors(3)[0]     # zero literal: empty grade set
    assert z.specialize(3).eval().to_dict() == {}


def test_ieee_specials_propagate():
    """eval.rs:107-108: 1/0 -> inf, sqrt(<0) -> NaN, no status."""
    z = ga.mv(ga.GradeMapMV({0: [0.0]}, dim=3)) + ga.mv(ga.GradeMapMV({0: [0.0]}, dim=3))
    assert np.isinf(z.sinv().specialize(3).eval().to_dict()[0][0])
    m = ga.mv(ga.GradeMapMV({0: [-4.0]}, dim=3)) + ga.mv(ga.GradeMapMV({0: [0.0]}, dim=3))
    assert np.isnan(m.sqrt().specialize(3).eval().to_dict()[0][0])
