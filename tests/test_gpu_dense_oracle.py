"""Dense geometric products against the ORACLE ITSELF at the sizes the dense kernels are quoted on (n = 10 ... 13):
every (kernel, block size, waves per item) instantiation runtime.hip:prepare_step can pick is named here by a test that
compares with oracle_eval_batch.  The oracle builds the reference's 4^n-entry table (940 MB at n = 12), so the batches are
1-3 items; n = 13 restricts the LEFT operand to grades 0..5 (the leading 29 % of the table: 1.1 GB instead of 3.8 GB), which
keeps the product on the dense kernels (>= 1/8 of the full table).

Tolerance: |got - want| <= 4 eps sum|terms| per component (eps of the program's dtype), the bound DESIGN.md states.
"""
import numpy as np
import pytest

import gaast_amd as ga
from helpers import abs_terms_bound, bits_to_row, full_grades, hip_eval_batch, oracle_eval_batch, row_to_bits, rows_of

pytestmark = pytest.mark.gpu


def _gp(n, left_grades=None):
    lg = full_grades(n) if left_grades is None else left_grades

    def build(B):
        return B.input(0, lg, n) * B.input(1, full_grades(n), n)
    return build, lg


def _check(n, metric, left_grades, variants, batch, seed, out_grades=None):
    """variants: [(dtype, flags, expected launch-name prefix)]; one oracle evaluation serves them all"""
    build, lg = _gp(n, left_grades)
    rng = np.random.default_rng(seed)
    rows = {0: rows_of(n, lg, batch, rng, np.float32), 1: rows_of(n, full_grades(n), batch, rng, np.float32)}
    rows64 = {s: r.astype(np.float64) for s, r in rows.items()}     # the same values, exactly representable in both types
    alg = metric if any(m != 1.0 for m in metric) else n
    want, wmask = oracle_eval_batch(build, alg, rows64, batch)
    og = [k for k in range(n + 1) if (wmask >> k) & 1]
    for dtype, flags, prefix in variants:
        got, mask, spec = hip_eval_batch(build, alg, rows if dtype == ga.F32 else rows64, batch, dtype=dtype, flags=flags)
        assert mask == wmask
        assert any(l.startswith(prefix) for l in spec.launches()), (prefix, spec.launches())
        eps = 2.0 ** -23 if dtype == ga.F32 else 2.0 ** -52
        for i in range(batch):
            S = abs_terms_bound(n, row_to_bits(n, lg, rows64[0][i]), row_to_bits(n, full_grades(n), rows64[1][i]))
            bound = 4 * eps * bits_to_row(n, og, S) + 1e-300
            err = np.abs(got[i].astype(np.float64) - want[i])
            assert np.all(err <= bound), (prefix, dtype, i, float((err / bound).max()))
        assert np.abs(want).max() > 1.0     # a real product, not zeros


def test_n10_f64_vector_kernel():
    """k_gp_dense<double, false, 256>: 64 lanes per item, four items per workgroup"""
    _check(10, [1.0] * 10, None, [(ga.F64, 0, "product_dense[")], batch=5, seed=10)


def test_n11_f32_two_waves_per_item_and_f64():
    """k_gp_mfma32<false, 256> with 2 waves per item (2 items per workgroup, ragged last workgroup); k_gp_dense f32 / f64"""
    _check(11, [1.0] * 11, None, [(ga.F32, 0, "product_dense_mfma["), (ga.F32, ga.FLAG_NO_MFMA, "product_dense["),
                                  (ga.F64, 0, "product_dense[")], batch=3, seed=11)


def test_n12_f32_and_f64_against_the_oracle():
    """BASELINE configs[2] kernel (k_gp_mfma32<false, 256>, 4 waves per item) and the f64 workload r12d
    (k_gp_dense<double, false, 256>), two items, all 4096 components, against the reference's 16.7 M-entry table"""
    _check(12, [1.0] * 12, None, [(ga.F32, 0, "product_dense_mfma["), (ga.F32, ga.FLAG_NO_MFMA, "product_dense["),
                                  (ga.F64, 0, "product_dense[")], batch=2, seed=12)


def test_n12_mixed_signature_negative_vectors_among_lo_and_hi_bits():
    """-1 at positions 1, 3 (lo bits of every kernel), 7, 10 (hi bits): the matrix-core kernel takes the lo signs as lane
    constants; the vector kernel needs four like-signed lo vectors and gets them by a basis permutation"""
    metric = [1.0, -1.0, 1.0, -1.0, 1.0, 1.0, 1.0, -1.0, 1.0, 1.0, -1.0, 1.0]
    _check(12, metric, None, [(ga.F32, 0, "product_dense_mfma["), (ga.F64, 0, "product_dense[")], batch=1, seed=13)


def test_n12_degenerate_metric_null_vector_first():
    """the reference's PGA habit of putting the null vector first (eval.rs:132), at n = 12, plus a -1 and a second null
    vector: the basis permutation moves the null vectors to the hi bits"""
    metric = [0.0, 1.0, 1.0, 1.0, 1.0, -1.0, 1.0, 1.0, 1.0, 0.0, 1.0, 1.0]
    _check(12, metric, None, [(ga.F32, 0, "product_dense_mfma["), (ga.F64, 0, "product_dense[")], batch=1, seed=14)


def test_n13_eight_waves_per_item_512_thread_kernels():
    """k_gp_mfma32<false, 512> (8 waves per item), k_gp_dense<float, false, 512>, k_gp_dense<double, false, 512>"""
    _check(13, [1.0] * 13, [0, 1, 2, 3, 4, 5], [(ga.F32, 0, "product_dense_mfma["), (ga.F32, ga.FLAG_NO_MFMA, "product_dense["),
                                                (ga.F64, 0, "product_dense[")], batch=1, seed=15)


@pytest.mark.parametrize("metric", [[0.0] + [1.0] * 7, [-1.0] + [1.0] * 7, [-1.0] * 8, [1.0, 0.0, -1.0, 1.0, 0.0, -1.0, 1.0, -1.0],
                                    [-1.0, -1.0, 1.0, 0.0, -1.0, -1.0, 1.0, 1.0, -1.0]])
def test_n8_n9_any_diagonal_metric_stays_on_the_dense_kernels(metric):
    """PGA-style null vector first, STA-style time first, Cl(0,8), and mixtures: k_gp_mfma16 in f32 (lane-constant lo
    signs), k_gp_dense in f64 (permutation to four like-signed lo vectors; NEGLO instantiation for Cl(0,8))"""
    n = len(metric)
    variants = [(ga.F32, 0, "product_dense_mfma[")]
    if sum(m == 1.0 for m in metric) >= 4 or sum(m == -1.0 for m in metric) >= 4:
        variants += [(ga.F64, 0, "product_dense["), (ga.F32, ga.FLAG_NO_MFMA, "product_dense[")]
    _check(n, metric, None, variants, batch=9, seed=80 + n)


def test_dense_kernels_need_enough_non_null_vectors():
    """six null vectors out of eight leave fewer than four non-null lo candidates: the product stays on the exact list
    kernels (still correct, bit-exact)"""
    metric = [0.0, 0.0, 1.0, 0.0, 0.0, -1.0, 0.0, 0.0]
    n, batch = 8, 3
    build, lg = _gp(n)
    rng = np.random.default_rng(5)
    rows = {0: rows_of(n, lg, batch, rng), 1: rows_of(n, lg, batch, rng)}
    want, _ = oracle_eval_batch(build, metric, rows, batch)
    got, _, spec = hip_eval_batch(build, metric, rows, batch)
    assert not any("product_dense" in l for l in spec.launches()), spec.launches()
    assert np.array_equal(got, want)
