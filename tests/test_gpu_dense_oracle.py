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
from oracle import pyoracle as ogm
from helpers import abs_terms_bound, bits_to_row, full_grades, hip_eval_batch, oracle_eval_batch, row_to_bits, rows_of

pytestmark = pytest.mark.gpu


def _gp(n, left_grades=None, right_grades=None):
    lg = full_grades(n) if left_grades is None else left_grades
    rg = full_grades(n) if right_grades is None else right_grades

    def build(B):
        return B.input(0, lg, n) * B.input(1, rg, n)
    return build, lg


def _check(n, metric, left_grades, variants, batch, seed, out_grades=None, eps_factor=4, exact_order_too=False, right_grades=None,
           label_has=None):
    """variants: [(dtype, flags, expected launch-name prefix, HIP kernel)]; one oracle evaluation serves them all.  The launch
    label names the HIP kernel instantiation prepare_step picked ("<what> :: <kernel<...>>", the name rocprofv3 reports):
    the test asserts it, so a docstring cannot go stale about which kernel it covers."""
    build, lg = _gp(n, left_grades, right_grades)
    rg = full_grades(n) if right_grades is None else right_grades
    rng = np.random.default_rng(seed)
    rows = {0: rows_of(n, lg, batch, rng, np.float32), 1: rows_of(n, rg, batch, rng, np.float32)}
    rows64 = {s: r.astype(np.float64) for s, r in rows.items()}     # the same values, exactly representable in both types
    alg = metric if any(m != 1.0 for m in metric) else n
    want, wmask = oracle_eval_batch(build, alg, rows64, batch)
    og = [k for k in range(n + 1) if (wmask >> k) & 1]
    for dtype, flags, prefix, kernel in variants:
        got, mask, spec = hip_eval_batch(build, alg, rows if dtype == ga.F32 else rows64, batch, dtype=dtype, flags=flags)
        assert mask == wmask
        assert any(l.startswith(prefix) and l.split(" :: ")[-1].startswith(kernel) for l in spec.launches()), (prefix, kernel, spec.launches())
        if label_has:
            assert any(label_has in l for l in spec.launches()), (label_has, spec.launches())
        eps = 2.0 ** -23 if dtype == ga.F32 else 2.0 ** -52
        general = any(m not in (1.0, -1.0, 0.0) for m in metric)
        for i in range(batch):
            S = abs_terms_bound(n, row_to_bits(n, lg, rows64[0][i]), row_to_bits(n, rg, rows64[1][i]), metric if general else None)
            bound = eps_factor * eps * bits_to_row(n, og, S) + 1e-300
            err = np.abs(got[i].astype(np.float64) - want[i])
            assert np.all(err <= bound), (prefix, dtype, i, float((err / bound).max()))
        assert np.abs(want).max() > 1.0     # a real product, not zeros
    if exact_order_too:     # the reference's summation order stays available, bit for bit (k_product_csr: general coefficients)
        got, mask, spec = hip_eval_batch(build, alg, rows64, batch, dtype=ga.F64, flags=ga.FLAG_EXACT_ORDER)
        assert any("product_csr" in l for l in spec.launches()), spec.launches()
        assert np.array_equal(got, want)


def test_n10_f64_matrix_core_and_vector_kernels():
    """k_gp_mfma16x4<double, false, 10> (four waves per item) and, behind GAAST_FLAG_NO_MFMA, k_gp_dense<double, false, 256>:
    64 lanes per item, four items per workgroup"""
    _check(10, [1.0] * 10, None, [(ga.F64, 0, "product_dense_mfma[", "k_gp_mfma16x4<double,false,10,"),
                                  (ga.F64, ga.FLAG_NO_MFMA, "product_dense[", "k_gp_dense<double,false,256,false>")], batch=5, seed=10)


def test_n11_f32_two_waves_per_item_and_f64():
    """k_gp_mfma32p<false, 11> with 2 waves per item (2 items per workgroup, ragged last workgroup); k_gp_mfma16x4<double>
    with 8 waves per item; k_gp_dense f32 / f64"""
    _check(11, [1.0] * 11, None, [(ga.F32, 0, "product_dense_mfma[", "k_gp_mfma32p<false,11>"),
                                  (ga.F32, ga.FLAG_NO_MFMA, "product_dense[", "k_gp_dense<float,false,256,false>"),
                                  (ga.F64, 0, "product_dense_mfma[", "k_gp_mfma16x4<double,false,11,"),
                                  (ga.F64, ga.FLAG_NO_MFMA, "product_dense[", "k_gp_dense<double,false,256,false>")], batch=3, seed=11)


def test_n12_f32_and_f64_against_the_oracle():
    """BASELINE configs[2] kernel (k_gp_mfma32p<false, 12>, 4 waves per item) and the f64 workload r12d
    (k_gp_mfma16x4<double, false, 12>, 16 waves per item), the vector kernels behind GAAST_FLAG_NO_MFMA: two items, all 4096
    components, against the reference's 16.7 M-entry table"""
    _check(12, [1.0] * 12, None, [(ga.F32, 0, "product_dense_mfma[", "k_gp_mfma32p<false,12>"),
                                  (ga.F32, ga.FLAG_NO_MFMA, "product_dense[", "k_gp_dense<float,false,256,false>"),
                                  (ga.F64, 0, "product_dense_mfma[", "k_gp_mfma16x4<double,false,12,"),
                                  (ga.F64, ga.FLAG_NO_MFMA, "product_dense[", "k_gp_dense<double,false,256,false>")], batch=2, seed=12)


def test_n12_mixed_signature_negative_vectors_among_lo_and_hi_bits():
    """-1 at positions 1, 3 (lo bits of every kernel), 7, 10 (hi bits): the matrix-core kernel takes the lo signs as lane
    constants; the vector kernel needs four like-signed lo vectors and gets them by a basis permutation"""
    metric = [1.0, -1.0, 1.0, -1.0, 1.0, 1.0, 1.0, -1.0, 1.0, 1.0, -1.0, 1.0]
    _check(12, metric, None, [(ga.F32, 0, "product_dense_mfma[", "k_gp_mfma32p<false,12>"), (ga.F64, 0, "product_dense_mfma[", "k_gp_mfma16x4<double,false,12,"),
                              (ga.F64, ga.FLAG_NO_MFMA, "product_dense[", "k_gp_dense<double,false,256,")], batch=1, seed=13)


def test_n12_degenerate_metric_null_vector_first():
    """the reference's PGA habit of putting the null vector first (eval.rs:132), at n = 12, plus a -1 and a second null
    vector: the basis permutation moves the null vectors to the hi bits"""
    metric = [0.0, 1.0, 1.0, 1.0, 1.0, -1.0, 1.0, 1.0, 1.0, 0.0, 1.0, 1.0]
    _check(12, metric, None, [(ga.F32, 0, "product_dense_mfma[", "k_gp_mfma32p<true,12>"), (ga.F64, 0, "product_dense_mfma[", "k_gp_mfma16x4<double,true,12,"),
                              (ga.F64, ga.FLAG_NO_MFMA, "product_dense[", "k_gp_dense<double,true,256,")], batch=1, seed=14)


@pytest.mark.parametrize("n", [8, 10, 11])
def test_partial_left_operand_takes_the_general_staging(n):
    """a left operand that holds only some grades (zeros elsewhere, no 16-byte-piece fast path): k_gp_mfma16x4<float>
    (n = 8), k_gp_mfma32p (n = 10, 11) and k_gp_mfma16x4<double> (1, 4, 8 waves per item) through their general staging"""
    lg = [0, 1, 2, 3, 4, 5, 6] if n == 8 else [0, 1, 2, 3, 4, 5, 6, 7]   # enough entries for the dense kernels (>= 4^n / 8)
    f32_kernel = "k_gp_mfma16x4<float,false,8," if n == 8 else f"k_gp_mfma32p<false,{n}>"
    _check(n, [1.0] * n, lg, [(ga.F32, 0, "product_dense_mfma[", f32_kernel), (ga.F64, 0, "product_dense_mfma[", f"k_gp_mfma16x4<double,false,{n},")], batch=3, seed=40 + n)


def test_n13_eight_waves_per_item_512_thread_kernels():
    """k_gp_mfma32p<false, 13> (512 threads, 8 waves per item), k_gp_dense<float, false, 512>, k_gp_dense<double, false, 512>"""
    _check(13, [1.0] * 13, [0, 1, 2, 3, 4, 5], [(ga.F32, 0, "product_dense_mfma[", "k_gp_mfma32p<false,13>"),
                                                (ga.F32, ga.FLAG_NO_MFMA, "product_dense[", "k_gp_dense<float,false,512,false>"),
                                                (ga.F64, 0, "product_dense[", "k_gp_dense<double,false,512,false>")], batch=1, seed=15)


@pytest.mark.parametrize("metric", [[0.0] + [1.0] * 7, [-1.0] + [1.0] * 7, [-1.0] * 8, [1.0, 0.0, -1.0, 1.0, 0.0, -1.0, 1.0, -1.0],
                                    [-1.0, -1.0, 1.0, 0.0, -1.0, -1.0, 1.0, 1.0, -1.0]])
def test_n8_n9_any_diagonal_metric_stays_on_the_dense_kernels(metric):
    """PGA-style null vector first, STA-style time first, Cl(0,8), and mixtures: k_gp_mfma16x4<float> and k_gp_mfma16x4<double>
    (lane-constant lo signs), k_gp_dense in both types behind GAAST_FLAG_NO_MFMA (permutation to four like-signed lo
    vectors; NEGLO instantiation for Cl(0,8))"""
    n = len(metric)
    dg = "true" if 0.0 in metric else "false"
    variants = [(ga.F32, 0, "product_dense_mfma[", f"k_gp_mfma16x4<float,{dg},{n},"), (ga.F64, 0, "product_dense_mfma[", f"k_gp_mfma16x4<double,{dg},{n},")]
    if sum(m == 1.0 for m in metric) >= 4 or sum(m == -1.0 for m in metric) >= 4:
        neglo = "true" if sum(m == 1.0 for m in metric) < 4 else "false"
        variants += [(ga.F64, ga.FLAG_NO_MFMA, "product_dense[", f"k_gp_dense<double,{dg},256,{neglo}>"),
                     (ga.F32, ga.FLAG_NO_MFMA, "product_dense[", f"k_gp_dense<float,{dg},256,{neglo}>")]
    _check(n, metric, None, variants, batch=9, seed=80 + n)


@pytest.mark.parametrize("chunk", range(4))
def test_random_dense_products_on_the_default_path(chunk):
    """Randomised: n = 8, 9, 10, both value types, random +-1 / 0 metrics (at least five non-null vectors), operands that miss
    one or two grades, a random projection of the result, a shared (batch-1) operand now and then -- the matrix-core kernels'
    maps, basis permutation, general staging and result maps against the oracle, within 4 eps sum |terms|"""
    rng = np.random.default_rng(7000 + chunk)
    dense_cases = 0
    for case in range(6):
        n = int(rng.choice([8, 9, 10]))
        dtype = ga.F32 if rng.random() < 0.5 else ga.F64
        while True:
            metric = [float(x) for x in rng.choice([1.0, -1.0, 0.0], size=n, p=[0.5, 0.35, 0.15])]
            if sum(m != 0.0 for m in metric) >= 5:
                break
        allg = list(range(n + 1))
        lg = sorted(set(allg) - set(int(g) for g in rng.choice(allg, size=int(rng.integers(0, 2)), replace=False)))
        rg = sorted(set(allg) - set(int(g) for g in rng.choice(allg, size=int(rng.integers(0, 2)), replace=False)))
        og_sel = sorted(set(allg) - set(int(g) for g in rng.choice(allg, size=int(rng.integers(0, 3)), replace=False)))
        batch = int(rng.integers(1, 12))
        shared = rng.random() < 0.3
        build = lambda B, lg=lg, rg=rg, og_sel=og_sel, n=n: (B.input(0, lg, n) * B.input(1, rg, n)).gselect(og_sel)
        rows = {0: rows_of(n, lg, 1 if shared else batch, rng, np.float32), 1: rows_of(n, rg, batch, rng, np.float32)}
        rows64 = {s_: r.astype(np.float64) for s_, r in rows.items()}
        want, wmask = oracle_eval_batch(build, metric, rows64, batch)
        got, mask, spec = hip_eval_batch(build, metric, rows if dtype == ga.F32 else rows64, batch, dtype=dtype)
        where = (n, dtype, metric, lg, rg, og_sel, batch, shared, spec.launches())
        assert mask == wmask, where
        if not any("product_dense" in l for l in spec.launches()):
            # exact list kernels: bit for bit, in f32 against the oracle's F32 mode (same statements on binary32 values)
            exact = want if dtype == ga.F64 else oracle_eval_batch(build, metric, rows64, batch, mode=ogm.EVAL_F32)[0]
            assert np.array_equal(got.astype(np.float64), exact), where
            continue
        dense_cases += 1
        og = [k for k in range(n + 1) if (wmask >> k) & 1]
        eps = 2.0 ** -23 if dtype == ga.F32 else 2.0 ** -52
        for i in range(batch):
            S = abs_terms_bound(n, row_to_bits(n, lg, rows64[0][0 if shared else i]), row_to_bits(n, rg, rows64[1][i]))
            bound = 4 * eps * bits_to_row(n, og, S) + 1e-300
            err = np.abs(got[i].astype(np.float64) - want[i])
            assert np.all(err <= bound), (where, i, float((err / bound).max()))
    assert dense_cases >= 4, dense_cases     # the draw is meant to land on the dense kernels


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


def test_n14_sixteen_waves_per_item_against_a_sparse_left_operand():
    """k_gp_mfma32<false, 1024>: n = 14 (16,384 components, 128 KiB of LDS per item, 16 waves).  The reference's table
    would be 15 GB, so the check is the bitmask form C[a ^ b] += s(a, b) A[a] B[b] with a SPARSE left operand (48 non-zero
    components spread over all grades) and a dense right one -- every output component, every block sign -- plus basis
    blades exactly.  Mixed signature: -1 among lo and hi bits."""
    from helpers import _POP16, blades_in_row_order
    n, N, batch = 14, 1 << 14, 2
    metric = [1.0, -1.0, 1.0, 1.0, 1.0, 1.0, -1.0, 1.0, 1.0, 1.0, 1.0, -1.0, 1.0, 1.0]
    rng = np.random.default_rng(14)
    blades = blades_in_row_order(n, full_grades(n))
    pos_of = np.zeros(N, dtype=np.int64)
    pos_of[blades] = np.arange(N)
    A = np.zeros((batch, N), np.float32)
    nz = [rng.choice(N, 48, replace=False) for _ in range(batch)]
    for i in range(batch):
        A[i, nz[i]] = rng.uniform(-1, 1, 48).astype(np.float32)
    B = rng.uniform(-1, 1, (batch, N)).astype(np.float32)
    build, _ = _gp(n)
    got, mask, spec = hip_eval_batch(build, metric, {0: A, 1: B}, batch, dtype=ga.F32)
    assert any(l.startswith("product_dense_mfma[gp n=14") and l.endswith(":: k_gp_mfma32<false,1024>") for l in spec.launches()), spec.launches()
    neg = sum(1 << i for i, g in enumerate(metric) if g < 0)
    bmask = np.arange(N, dtype=np.int64)
    for i in range(batch):
        want = np.zeros(N)
        absum = np.zeros(N)
        Bbits = np.zeros(N)
        Bbits[blades] = B[i].astype(np.float64)
        for p in nz[i]:
            a = int(blades[p])
            par = np.zeros(N, dtype=np.int64)
            for sft in range(1, n):
                par += _POP16[(a >> sft) & bmask]
            par += _POP16[a & bmask & neg]
            term = np.where(par & 1, -1.0, 1.0) * float(A[i, p]) * Bbits
            np.add.at(want, a ^ bmask, term)
            np.add.at(absum, a ^ bmask, np.abs(term))
        err = np.abs(got[i].astype(np.float64) - want[blades])
        assert np.all(err <= 4 * 2.0 ** -23 * absum[blades] + 1e-30), float((err / (4 * 2.0 ** -23 * absum[blades] + 1e-30)).max())
    # basis blades: e_S e_T = +- e_{S ^ T}, exactly
    a_idx, b_idx = rng.integers(0, N, 8), rng.integers(0, N, 8)
    ra, rb = np.zeros((8, N), np.float32), np.zeros((8, N), np.float32)
    ra[np.arange(8), a_idx] = 1.0
    rb[np.arange(8), b_idx] = 1.0
    got, _, _ = hip_eval_batch(build, metric, {0: ra, 1: rb}, 8, dtype=ga.F32)
    alg = ga.MetricAlgebra(metric)
    for i in range(8):
        res, coeff = alg.ortho_basis_blades_gp(int(blades[a_idx[i]]), int(blades[b_idx[i]]))
        want = np.zeros(N, np.float32)
        want[pos_of[res]] = coeff
        assert np.array_equal(got[i], want), i


@pytest.mark.parametrize("n,dtype", [(7, ga.F32), (7, ga.F64), (8, ga.F32), (8, ga.F64), (10, ga.F64), (10, ga.F32)])
def test_negative_zero_operands_leave_no_trace(n, dtype):
    """The FAST staging of k_gp_mfma16x4 (and of k_gp_mfma7, whose u = 1 tile elements also take a sign flip AFTER the accumulation: each
    component adds one unflipped element, so a zero sum still ends as +0.0) does not apply the `0.0 + x` of the reference's operand copy (graded.rs:74): a -0.0
    component can only contribute +-0 to sums that start from +0.0, so the results -- all-zero sums included -- must equal, BIT
    FOR BIT, those of the same rows with every -0.0 replaced by +0.0, and no result may be -0.0 where the reference has +0.0.
    Rows: random values with a third of the components set to -0.0, plus rows that are entirely +-0.0."""
    npdt = np.float32 if dtype == ga.F32 else np.float64
    rng = np.random.default_rng(99 + n)
    batch = 6
    build, lg = _gp(n)
    rows = {s: rows_of(n, lg, batch, rng, npdt) for s in range(2)}
    for s in range(2):
        holes = rng.random(rows[s].shape) < 0.33
        rows[s][holes] = -0.0
    rows[0][4, :] = -0.0                      # (-0) * B: every sum is a sum of zeros
    rows[1][5, :] = -0.0
    clean = {s: np.where(rows[s] == 0, npdt(0.0), rows[s]) for s in range(2)}
    got, _, spec = hip_eval_batch(build, n, rows, batch, dtype=dtype)
    ref, _, _ = hip_eval_batch(build, n, clean, batch, dtype=dtype)
    assert any("product_dense_mfma[" in l for l in spec.launches())
    assert np.array_equal(got, ref) and np.array_equal(np.signbit(got), np.signbit(ref))
    assert not np.signbit(got[4]).any() and not np.signbit(got[5]).any() and not got[4].any()


@pytest.mark.parametrize("metric", [[2.0, 0.5, -3.0, 1.0, 1.0, 1.0, 1.0, 1.0], [0.0, 7.5, -0.125, 1.0, 2.0, 2.0, -1.0, 3.0],
                                    [1.5, -2.0, 0.3, 1.0, -1.0, 4.0, 1.0, 0.0, 2.5]])
def test_general_diagonal_metrics_run_on_the_dense_kernels(metric):
    """algebra.rs:148-165 (`impl MetricAlgebra for [f64; D]`) multiplies by ANY base_vec_dot (:79-81).  Round 2 sent every
    entry other than +-1 / 0 to k_product_csr (2.2 M products/s at n = 8 against ~800 M/s).  Now the dense kernels run in the
    rescaled basis f_i = e_i / sqrt|g_i|: operands times w_S while staged, result times 1 / w_T when stored (plan.cpp:
    dense_scales_ok / blade_scale).  Bound: 8 eps sum|terms| per component -- the 4 eps of the +-1 case plus three roundings
    per term (w_S A_S, w_U B_U, C'_T / w_T) and the roundings of the three factors themselves.  With GAAST_FLAG_EXACT_ORDER the
    product stays on k_product_csr and is bit-exact."""
    n = len(metric)
    dg = "true" if 0.0 in metric else "false"
    variants = [(ga.F32, 0, "product_dense_mfma[", f"k_gp_mfma16x4<float,{dg},{n},"), (ga.F64, 0, "product_dense_mfma[", f"k_gp_mfma16x4<double,{dg},{n},"),
                (ga.F64, ga.FLAG_NO_MFMA, "product_dense[", f"k_gp_dense<double,{dg},256,"), (ga.F32, ga.FLAG_NO_MFMA, "product_dense[", f"k_gp_dense<float,{dg},256,")]
    _check(n, metric, None, variants, batch=7, seed=300 + n, eps_factor=8, exact_order_too=True)


def test_general_diagonal_metric_at_n12_and_partial_operands():
    """a 12-dimensional general metric on the headline kernel (k_gp_mfma32p) and on k_gp_mfma16x4<double>, and at n = 10 with a
    left operand that misses grades (the scale table follows the map, not the blade index)"""
    metric = [1.0, 2.0, -0.5, 1.0, 3.0, 1.0, -1.0, 0.25, 1.0, 1.0, -4.0, 1.5]
    _check(12, metric, None, [(ga.F32, 0, "product_dense_mfma[", "k_gp_mfma32p<false,12,true>"), (ga.F64, 0, "product_dense_mfma[", "k_gp_mfma16x4<double,false,12,")],
           batch=1, seed=312, eps_factor=8)
    metric10 = [2.0, 1.0, -0.5, 1.0, 0.0, 1.0, -1.0, 0.25, 3.0, 1.0]
    _check(10, metric10, [0, 1, 2, 3, 4, 5, 6, 7], [(ga.F32, 0, "product_dense_mfma[", "k_gp_mfma32p<true,10,true>"),
                                                     (ga.F64, 0, "product_dense_mfma[", "k_gp_mfma16x4<double,true,10,")], batch=2, seed=310, eps_factor=8)


def test_a_metric_too_wild_to_rescale_stays_on_the_list_kernels():
    """1e200 and 1e-200 among the squares: w_S would leave the range in which the rescaling is harmless (2^+-300 in f64,
    2^+-40 in f32: plan.cpp dense_scales_ok) -> the exact kernels, bit-exact"""
    metric = [1e200, 1e-200, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0]
    build, lg = _gp(8)
    rng = np.random.default_rng(5)
    rows = {0: rows_of(8, lg, 2, rng), 1: rows_of(8, lg, 2, rng)}
    want, _ = oracle_eval_batch(build, metric, rows, 2)
    got, _, spec = hip_eval_batch(build, metric, rows, 2)
    assert not any("product_dense" in l for l in spec.launches()), spec.launches()
    assert np.array_equal(got, want)


EVEN = lambda n: [k for k in range(n + 1) if k % 2 == 0]
ODD = lambda n: [k for k in range(n + 1) if k % 2 == 1]


@pytest.mark.parametrize("lpar,rpar", [("even", "even"), ("even", "odd"), ("odd", "even"), ("odd", "odd")])
@pytest.mark.parametrize("n,metric", [(9, [1.0] * 9), (10, [1.0, -1.0, 1.0, 1.0, -1.0, 1.0, 0.0, 1.0, -1.0, 1.0]), (12, [1.0] * 12)])
def test_parity_pure_operands_run_in_the_even_subalgebra(n, metric, lpar, rpar):
    """The reference only multiplies the entries it needs (specialize.rs:162-183): even x even (rotor composition, the second
    product of every sandwich) is a quarter of the 4^n table.  Round 2's dense kernels always did 4^n multiply-adds; now a
    product of parity-pure operands is ONE product in Cl(n - 1) (plan.cpp: parity_reduced_frame) -- 4^(n-1) multiply-adds --
    for all four parity combinations, against the oracle within 4 eps sum|terms| in both value types."""
    if n == 12 and (lpar, rpar) not in (("even", "even"), ("odd", "even")):
        pytest.skip("n = 12: two of the four cases (each builds a 4.2 M-entry oracle table)")
    lg, rg = (EVEN if lpar == "even" else ODD)(n), (EVEN if rpar == "even" else ODD)(n)
    dg = "true" if 0.0 in metric else "false"
    f32_kernel = {8: f"k_gp_mfma16x4<float,{dg},8,", 9: f"k_gp_mfma16x4<float,{dg},9,", 11: f"k_gp_mfma32p<{dg},11>"}[n - 1]
    variants = [(ga.F32, 0, "product_dense_mfma[", f32_kernel), (ga.F64, 0, "product_dense_mfma[", f"k_gp_mfma16x4<double,{dg},{n - 1},")]
    _check(n, metric, lg, variants, batch=3 if n < 12 else 1, seed=500 + n, right_grades=rg, label_has=f"{lpar} x {rpar} in Cl({n - 1})")


def test_parity_pure_with_a_general_metric_and_partial_grades():
    """even x even with a general diagonal metric (the reduction's own factors are powers of 1 / g_p: they ride in the scale
    tables), and a rotor-like left operand holding only grades 0, 2, 4 of the even grades (general staging, zeros elsewhere)"""
    metric = [2.0, 1.0, -0.5, 1.0, 3.0, 1.0, -1.0, 0.25, 1.5]
    _check(9, metric, EVEN(9), [(ga.F32, 0, "product_dense_mfma[", "k_gp_mfma16x4<float,false,8,"), (ga.F64, 0, "product_dense_mfma[", "k_gp_mfma16x4<double,false,8,")],
           batch=3, seed=520, right_grades=ODD(9), eps_factor=8, exact_order_too=True, label_has="even x odd in Cl(8)")
    _check(10, [1.0] * 10, [0, 2, 4, 6], [(ga.F64, 0, "product_dense_mfma[", "k_gp_mfma16x4<double,false,9,")], batch=2, seed=521,
           right_grades=EVEN(10), label_has="even x even in Cl(9)")


@pytest.mark.parametrize("n,metric", [(8, [1.0] * 8), (9, [1.0] * 6 + [-1.0] * 3), (10, [1.0] * 10), (9, [2.0, 1.0, -0.5, 1.0, 1.0, 3.0, -1.0, 1.0, 0.25]),
                                      (8, [1.0, 1.0, 1.0, 0.0, 1.0, -1.0, 1.0, 1.0])])   # (a null vector: rows of different lengths -- the list stays CSR, not in registers)
def test_rotor_sandwich_beyond_the_fused_slab_is_one_launch(n, metric):
    """BASELINE configs[4]'s pipeline R X ~R (eval.rs:61-86 with the cached operand R X, README.md:62-67) where it no longer fits
    a fused small-program kernel: the sparse product R X (n 2^(n-1) entries) is evaluated in the LDS staging of the dense
    product (R X) ~R (odd x even: Cl(n - 1)) -- ONE launch, the intermediate never goes through HBM.  The list keeps the
    reference's order and roundings, so the result is bit-identical to the two-launch plan (GAAST_FLAG_DEBUG_NO_CHAIN), and
    within 4 eps sum|terms| of the second product (8 eps: general metric) of the oracle; GAAST_FLAG_EXACT_ORDER: bit-exact."""
    even = [k for k in range(n + 1) if k % 2 == 0]
    build = lambda B: (lambda r, x: r * x * r.rev())(B.input(0, even, n), B.input(1, [1], n))
    batch = 5
    rng = np.random.default_rng(600 + n)
    rows = {0: rows_of(n, even, batch, rng), 1: rows_of(n, [1], batch, rng)}
    alg = metric if any(m != 1.0 for m in metric) else n
    want, wmask = oracle_eval_batch(build, alg, rows, batch)
    rx, rxmask = oracle_eval_batch(lambda B: B.input(0, even, n) * B.input(1, [1], n), alg, rows, batch)
    got, mask, spec = hip_eval_batch(build, alg, rows, batch)
    assert mask == wmask
    assert len(spec.launches()) == 1 and "product_dense" in spec.launches()[0] and "<- product_csr" in spec.launches()[0], spec.launches()
    two, _, spec2 = hip_eval_batch(build, alg, rows, batch, flags=ga.FLAG_DEBUG_NO_CHAIN)
    assert len(spec2.launches()) == 2, spec2.launches()
    assert np.array_equal(got, two)                      # the very bits of the two-launch plan
    general = any(m not in (1.0, -1.0, 0.0) for m in metric)
    odd = [k for k in range(n + 1) if (rxmask >> k) & 1]
    og = [k for k in range(n + 1) if (wmask >> k) & 1]
    for i in range(batch):
        S = abs_terms_bound(n, row_to_bits(n, odd, rx[i]), row_to_bits(n, even, rows[1 - 1][i]), metric if general else None)
        bound = (8 if general else 4) * 2.0 ** -52 * bits_to_row(n, og, S) + 1e-300
        err = np.abs(got[i] - want[i])
        assert np.all(err <= bound), (i, float((err / bound).max()))
    exact, _, spec3 = hip_eval_batch(build, alg, rows, batch, flags=ga.FLAG_EXACT_ORDER)
    assert np.array_equal(exact, want)


def _oracle_and_hip(build, alg, rows, batch, dtype=ga.F64, flags=0):
    want, wmask = oracle_eval_batch(build, alg, {s: r.astype(np.float64) for s, r in rows.items()}, batch)
    got, mask, spec = hip_eval_batch(build, alg, rows, batch, dtype=dtype, flags=flags)
    assert mask == wmask
    return want, got.astype(np.float64), spec


@pytest.mark.parametrize("n,dtype,right_grades,kernel", [
    (9, ga.F64, "even", "k_gp_mfma16x4<double,false,8,"),        # register-prefetch staging, the list reads OTHER rows than the right operand
    (9, ga.F64, "partial", "k_gp_mfma16x4<double,false,8,"),     # right operand misses a grade: general staging of a chained step
    (9, ga.F32, "even", "k_gp_mfma16x4<float,false,8,"),
    (8, ga.F32, "even", "k_gp_mfma7<float,"),              # Cl(7): one wave per item
    (8, ga.F64, "partial", "k_gp_mfma7<double,"),          # ... general staging of a chained step
    (11, ga.F32, "even", "k_gp_mfma32p<false,10,false,true>"),   # Cl(10): k_gp_mfma32p's general staging
])
def test_chained_products_with_unrelated_operands(n, dtype, right_grades, kernel):
    """(A x) B with three different inputs: the list's rows are NOT the dense product's right operand (the sandwich's shortcut
    does not apply), in both value types, on every kernel family a chained step can land on, with full and partial right
    operands.  One launch, bit-identical to the two-launch plan, within 4 eps sum|terms| of the oracle."""
    even = EVEN(n)
    rg = even if right_grades == "even" else [k for k in even if k != 4]
    build = lambda B: (B.input(0, even, n) * B.input(1, [1], n)) * B.input(2, rg, n)
    batch = 4 if n < 11 else 2
    npdt = np.float32 if dtype == ga.F32 else np.float64
    rng = np.random.default_rng(700 + n)
    rows = {0: rows_of(n, even, batch, rng, npdt), 1: rows_of(n, [1], batch, rng, npdt), 2: rows_of(n, rg, batch, rng, npdt)}
    want, got, spec = _oracle_and_hip(build, n, rows, batch, dtype)
    assert len(spec.launches()) == 1 and "<- product_" in spec.launches()[0] and kernel in spec.launches()[0], spec.launches()
    _, two, spec2 = _oracle_and_hip(build, n, rows, batch, dtype, flags=ga.FLAG_DEBUG_NO_CHAIN)
    assert len(spec2.launches()) == 2 and np.array_equal(got, two)
    ax, axmask = oracle_eval_batch(lambda B: B.input(0, even, n) * B.input(1, [1], n), n, {0: rows[0].astype(np.float64), 1: rows[1].astype(np.float64)}, batch)
    odd = [k for k in range(n + 1) if (axmask >> k) & 1]
    og = [k for k in range(n + 1) if k % 2 == 1]
    eps = 2.0 ** -23 if dtype == ga.F32 else 2.0 ** -52
    for i in range(batch):
        S = abs_terms_bound(n, row_to_bits(n, odd, ax[i]), row_to_bits(n, rg, rows[2][i].astype(np.float64)))
        # f32: the intermediate A x itself carries f32 roundings (n terms per component) on top of the second product's bound
        bound = (4 + (n if dtype == ga.F32 else 0)) * eps * bits_to_row(n, og, S) + 1e-300
        assert np.all(np.abs(got[i] - want[i]) <= bound), (i, float((np.abs(got[i] - want[i]) / bound).max()))


def test_parity_pure_products_accumulate_and_project():
    """a b + c d with four even operands (the second product ADDS into the buffer the first one wrote: beta = 1 in Cl(n - 1)), and
    (a b).g(2) + (a b).g(4)-style projections of a parity-pure product (a partial result map: general stores)"""
    n, batch = 9, 4
    even = EVEN(n)
    rng = np.random.default_rng(808)
    rows = {s: rows_of(n, even, batch, rng) for s in range(4)}
    build = lambda B: B.input(0, even, n) * B.input(1, even, n) + B.input(2, even, n) * B.input(3, even, n)
    want, got, spec = _oracle_and_hip(build, n, rows, batch)
    assert sum("even x even in Cl(8)" in l for l in spec.launches()) == 2, spec.launches()
    for i in range(batch):
        S = sum(abs_terms_bound(n, row_to_bits(n, even, rows[a][i]), row_to_bits(n, even, rows[b][i])) for a, b in ((0, 1), (2, 3)))
        bound = 5 * 2.0 ** -52 * bits_to_row(n, even, S) + 1e-300       # 4 eps per product, one more rounding for the sum
        assert np.all(np.abs(got[i] - want[i]) <= bound)
    build2 = lambda B: (B.input(0, even, n) * B.input(1, even, n)).g(2)
    rows2 = {0: rows[0], 1: rows[1]}
    want, got, spec = _oracle_and_hip(build2, n, rows2, batch)
    if any("product_dense" in l for l in spec.launches()):     # (the projection shrinks the list: the planner may keep it on a list kernel)
        for i in range(batch):
            S = abs_terms_bound(n, row_to_bits(n, even, rows[0][i]), row_to_bits(n, even, rows[1][i]))
            assert np.all(np.abs(got[i] - want[i]) <= 4 * 2.0 ** -52 * bits_to_row(n, [2], S) + 1e-300)
    else:
        assert np.array_equal(got, want)


@pytest.mark.parametrize("metric", [[1.0] * 7, [-1.0] * 7, [1.0, 1.0, 1.0, -1.0, 1.0, -1.0, -1.0], [0.0, 1.0, -1.0, 1.0, 0.0, 1.0, 1.0],
                                    [1.0, -1.0, 1.0, 1.0, -1.0, 0.0, 0.0], [0.0, 0.0, 1.0, 0.0, -1.0, 1.0, 0.0],
                                    [2.0, 0.5, -3.0, 1.0, 0.0, -0.25, 1.5]])
def test_n7_one_wave_per_item_on_the_matrix_cores(metric):
    """k_gp_mfma7<T> (both value types): the top basis vector split over the rows and columns of the 16 x 16 tile, the two halves
    of every component joined by a lane exchange.  Signatures with -1 / 0 in the lo bits' candidates (permuted basis), in the
    hi3 bits (zero words) and on the top vector (its square enters the result sign / drops the u = v = 1 half); a general metric
    (rescaled basis).  Full rows: register-prefetch staging and straight-line stores (MODE 2); behind GAAST_FLAG_NO_MFMA the
    vector kernel where it applies."""
    n = 7
    general = any(m not in (1.0, -1.0, 0.0) for m in metric)
    dg = "true" if 0.0 in metric else "false"
    variants = [(ga.F32, 0, "product_dense_mfma[", "k_gp_mfma7<float,"), (ga.F64, 0, "product_dense_mfma[", "k_gp_mfma7<double,")]
    if sum(m > 0 for m in metric) >= 4 or sum(m < 0 for m in metric) >= 4:
        variants += [(ga.F64, ga.FLAG_NO_MFMA, "product_dense[", f"k_gp_dense<double,{dg},256,"), (ga.F32, ga.FLAG_NO_MFMA, "product_dense[", f"k_gp_dense<float,{dg},256,")]
    _check(n, metric, None, variants, batch=70, seed=770, eps_factor=8 if general else 4, exact_order_too=general)


@pytest.mark.parametrize("left_grades,out_grades", [([0, 1, 2, 3, 4, 5], None), (None, [0, 2, 3, 5, 7]), ([1, 2, 3, 4, 5, 6, 7], [1, 2, 4, 6])])
def test_n7_general_staging_and_partial_results(left_grades, out_grades):
    """k_gp_mfma7 through its general staging (a left operand that misses grades: MODE 0) and its general stores (a projected
    result: MODE 1 with full operands), in a mixed signature with a permuted basis"""
    n = 7
    metric = [-1.0, 1.0, 1.0, -1.0, 1.0, 0.0, -1.0]
    lg = full_grades(n) if left_grades is None else left_grades
    og_sel = full_grades(n) if out_grades is None else out_grades
    build = lambda B: (B.input(0, lg, n) * B.input(1, full_grades(n), n)).gselect(og_sel)
    batch = 33
    for dtype in (ga.F32, ga.F64):
        rng = np.random.default_rng(771)
        rows = {0: rows_of(n, lg, batch, rng, np.float32), 1: rows_of(n, full_grades(n), batch, rng, np.float32)}
        rows64 = {s_: r.astype(np.float64) for s_, r in rows.items()}
        want, wmask = oracle_eval_batch(build, metric, rows64, batch)
        got, mask, spec = hip_eval_batch(build, metric, rows if dtype == ga.F32 else rows64, batch, dtype=dtype)
        assert mask == wmask
        assert any("k_gp_mfma7<" in l for l in spec.launches()), spec.launches()
        og = [k for k in range(n + 1) if (wmask >> k) & 1]
        eps = 2.0 ** -23 if dtype == ga.F32 else 2.0 ** -52
        for i in range(batch):
            S = abs_terms_bound(n, row_to_bits(n, lg, rows64[0][i]), row_to_bits(n, full_grades(n), rows64[1][i]))
            bound = 4 * eps * bits_to_row(n, og, S) + 1e-300
            err = np.abs(got[i].astype(np.float64) - want[i])
            assert np.all(err <= bound), (dtype, i, float((err / bound).max()))


def test_n7_products_accumulate():
    """a b + c d at n = 7: the second k_gp_mfma7 launch adds into the buffer the first one wrote (beta = 1: general stores)"""
    n, batch = 7, 19
    fg = full_grades(n)
    rng = np.random.default_rng(772)
    rows = {s_: rows_of(n, fg, batch, rng) for s_ in range(4)}
    build = lambda B: B.input(0, fg, n) * B.input(1, fg, n) + B.input(2, fg, n) * B.input(3, fg, n)
    want, got, spec = _oracle_and_hip(build, n, rows, batch)
    assert sum("k_gp_mfma7<double" in l for l in spec.launches()) == 2, spec.launches()
    for i in range(batch):
        S = sum(abs_terms_bound(n, row_to_bits(n, fg, rows[a][i]), row_to_bits(n, fg, rows[b][i])) for a, b in ((0, 1), (2, 3)))
        bound = 5 * 2.0 ** -52 * bits_to_row(n, fg, S) + 1e-300
        assert np.all(np.abs(got[i] - want[i]) <= bound), i


@pytest.mark.parametrize("lpar,rpar", [("even", "even"), ("even", "odd"), ("odd", "even"), ("odd", "odd")])
@pytest.mark.parametrize("metric", [[1.0] * 8, [1.0, -1.0, 1.0, 1.0, -1.0, 1.0, 0.0, -1.0]])
def test_n8_parity_pure_products_run_in_cl7_on_the_matrix_cores(lpar, rpar, metric):
    """rotor composition and the products of odd versors at n = 8: ONE product in Cl(7) on k_gp_mfma7 (round 3's first half ran
    them on the vector kernel at 1.8x the full product)"""
    n = 8
    grades = {"even": EVEN(n), "odd": ODD(n)}
    dg = "true" if 0.0 in metric else "false"
    _check(n, metric, grades[lpar], [(ga.F32, 0, "product_dense_mfma[", "k_gp_mfma7<float,"), (ga.F64, 0, "product_dense_mfma[", "k_gp_mfma7<double,")],
           batch=21, seed=780, right_grades=grades[rpar], label_has=f"{lpar} x {rpar} in Cl(7)")


@pytest.mark.parametrize("n,metric", [(8, [1.0] * 8), (9, [1.0] * 6 + [-1.0] * 3), (10, [1.0] * 10), (11, [1.0] * 11), (12, [1.0] * 4 + [-1.0] * 8)])
def test_projected_rotor_sandwich_is_one_launch_of_two_lists(n, metric):
    """(R X ~R).g(1), the rotor sandwich applied to a vector (README.md:62-67; eval.rs:61-86 with the cached R X), where the program
    no longer fits a fused small-program kernel: both products are lists (n 2^(n-1) entries each) and run in ONE
    k_product_ell_chain launch with the mid row R X in LDS -- reference order and roundings, so the result equals the oracle's
    and the two-launch plan's (GAAST_FLAG_DEBUG_NO_CHAIN) bit for bit.  The batch is not a multiple of the items a workgroup
    stages.  n = 8 would still fit the LDS interpreter (one launch as well): the chain kernel is preferred (11 % faster)."""
    even = EVEN(n)
    build = lambda B: (lambda r, x: (r * x * r.rev()).g(1))(B.input(0, even, n), B.input(1, [1], n))
    batch = 37 if n < 11 else 11 if n == 11 else 5
    rng = np.random.default_rng(900 + n)
    rows = {0: rows_of(n, even, batch, rng), 1: rows_of(n, [1], batch, rng)}
    alg = metric if any(m != 1.0 for m in metric) else n
    want, wmask = oracle_eval_batch(build, alg, rows, batch)
    # GAAST_FLAG_EXACT_ORDER: the reference's summation order, bit for bit
    got, mask, spec = hip_eval_batch(build, alg, rows, batch, flags=ga.FLAG_EXACT_ORDER)
    assert mask == wmask
    # the chain specialised per program through hiprtc (round 4) ...
    assert len(spec.launches()) == 1 and "<- product_" in spec.launches()[0] and "gaast_chain<double>" in spec.launches()[0], spec.launches()
    assert "re-ordered" not in spec.launches()[0]
    assert np.array_equal(got, want)
    # ... and the generic kernel it falls back to without run-time compilation (n = 8 without hiprtc: the LDS interpreter)
    gen, _, spec3 = hip_eval_batch(build, alg, rows, batch, flags=ga.FLAG_NO_JIT)
    assert len(spec3.launches()) == 1 and np.array_equal(gen, want), spec3.launches()
    if n > 8:
        assert "k_product_ell_chain<double>" in spec3.launches()[0], spec3.launches()
    two, _, spec2 = hip_eval_batch(build, alg, rows, batch, flags=ga.FLAG_DEBUG_NO_CHAIN | ga.FLAG_EXACT_ORDER)
    assert len(spec2.launches()) == (2 if n > 8 else 1) and np.array_equal(two, want), spec2.launches()    # (n = 8 without the chain: one fused launch)
    # DEFAULT (tolerance mode, like the dense products): the long rows of the second list are cut into slices summed by lanes of
    # their own -- "wavefront-parallel partial sums" -- within 4 eps sum |terms| of the oracle, per component
    tol, tmask, spec4 = hip_eval_batch(build, alg, rows, batch)
    assert tmask == wmask and len(spec4.launches()) == 1 and "gaast_chain<double>" in spec4.launches()[0], spec4.launches()
    if n <= 10:
        assert "re-ordered sums" in spec4.launches()[0], spec4.launches()
    # (round 4: when the rows' signs are balanced -- R^{6,3} at n = 9: 64 plus and 64 minus terms per slice -- list 2's terms are
    #  stored plus-first: the sign is the position, a term is two address additions and one fused multiply-add; the Euclidean n = 8, 10
    #  have all-plus rows and keep their sign words)
    if n <= 10:
        assert ("sign-sorted terms" in spec4.launches()[0]) == (n == 9), spec4.launches()
    mid, mmask = oracle_eval_batch(lambda B: B.input(0, even, n) * B.input(1, [1], n), alg, rows, batch)
    odd = [k for k in range(n + 1) if (mmask >> k) & 1]
    for i in range(batch):
        S = abs_terms_bound(n, row_to_bits(n, odd, mid[i]), row_to_bits(n, even, rows[0][i]))
        bound = 4 * 2.0 ** -52 * bits_to_row(n, [1], S) + 1e-300
        err = np.abs(tol[i] - want[i])
        assert np.all(err <= bound), (i, float((err / bound).max()))


@pytest.mark.parametrize("dtype", [ga.F64, ga.F32])
def test_list_chains_with_the_mid_row_on_either_side_and_unrelated_operands(dtype):
    """a (x b) with the mid row as the RIGHT operand of the second list; (a x) b with three different inputs (the second list stages a
    row of its own); the second list adding into a buffer another product wrote (beta = 1); in both value types, bit-exact against the
    oracle (f64: the reference's arithmetic; f32: the oracle's F32 mode, the same statements on binary32 values) AND the two-launch plan"""
    n, batch = 9, 21
    even = EVEN(n)
    npdt = np.float32 if dtype == ga.F32 else np.float64
    rng = np.random.default_rng(950)
    rows = {0: rows_of(n, even, batch, rng, npdt), 1: rows_of(n, [1], batch, rng, npdt), 2: rows_of(n, even, batch, rng, npdt), 3: rows_of(n, [1], batch, rng, npdt)}
    cases = [lambda B: (B.input(0, even, n) * (B.input(1, [1], n) * B.input(2, even, n))).g(1),
             lambda B: ((B.input(0, even, n) * B.input(1, [1], n)) * B.input(2, even, n)).g(1),
             lambda B: B.input(3, [1], n) + ((B.input(0, even, n) * B.input(1, [1], n)) * B.input(0, even, n).rev()).g(1)]
    all_rows = rows
    for k, build in enumerate(cases):
        rows = {s_: all_rows[s_] for s_ in ((0, 1, 2), (0, 1, 2), (0, 1, 3))[k]}
        got, mask, spec = hip_eval_batch(build, n, rows, batch, dtype=dtype, flags=ga.FLAG_EXACT_ORDER)
        two, _, spec2 = hip_eval_batch(build, n, rows, batch, dtype=dtype, flags=ga.FLAG_DEBUG_NO_CHAIN | ga.FLAG_EXACT_ORDER)
        assert np.array_equal(got, two), (k, spec.launches(), spec2.launches())
        if not any("ast_fused" in l or "ast_jit" in l for l in spec.launches()):     # (f32 slabs of the third program fit a fused launch: one launch anyway)
            assert any("gaast_chain<" in l for l in spec.launches()), (k, spec.launches())
            gen, _, spec3 = hip_eval_batch(build, n, rows, batch, dtype=dtype, flags=ga.FLAG_NO_JIT)    # the generic kernel
            assert np.array_equal(gen, two) and any("k_product_ell_chain<" in l or "ast_fused" in l for l in spec3.launches()), (k, spec3.launches())
            assert len(spec2.launches()) == len(spec.launches()) + 1, (k, spec.launches(), spec2.launches())
        rows64 = {s_: r.astype(np.float64) for s_, r in rows.items()}
        want, wmask = oracle_eval_batch(build, n, rows64, batch, mode=ogm.EVAL_RELEASE if dtype == ga.F64 else ogm.EVAL_F32)
        assert mask == wmask and np.array_equal(got.astype(np.float64), want), k


@pytest.mark.parametrize("metric", [[1.0] * 6,                                # R^6
                                    [1.0, 1.0, 1.0, 1.0, -1.0, -1.0],         # R^{4,2} (conformal space-time style)
                                    [1.0, -1.0, 1.0, -1.0, 1.0, -1.0],        # R^{3,3}: -1 in every bit group (lo, hi, top)
                                    [0.0, 1.0, 1.0, 1.0, -1.0, 0.0],          # null vectors in the lo AND the top group, a -1 in the top group
                                    [1.0, 0.0, -1.0, 0.0, 1.0, 1.0],          # null vectors in the lo and the hi group
                                    [-1.0] * 6,                               # Cl(0,6)
                                    [2.0, 0.5, -4.0, 1.0, -0.25, 3.0]])       # general diagonal metric: the rescaled basis
def test_n6_runs_on_the_matrix_cores_in_any_diagonal_metric(metric):
    """k_gp_mfma6<T> (round 4): n = 6 -- where PGA3D / CGA-adjacent algebras live -- on FOUR 16x16x4 instructions per item, both
    value types, any diagonal metric in the basis as it stands (signs and null vectors are slots of the operand images and bits
    of the accumulators; tools/proto/mfma6_tile.py is the CPU emulation of the decomposition).  Full operands, a ragged batch
    beyond the four items a wave keeps in flight, against the oracle within 4 eps sum |terms| (8 for the rescaled basis)."""
    general = any(m not in (1.0, -1.0, 0.0) for m in metric)
    sc = "true" if general else "false"
    variants = [(ga.F32, 0, "product_dense_mfma[", f"k_gp_mfma6<float,{sc},"), (ga.F64, 0, "product_dense_mfma[", f"k_gp_mfma6<double,{sc},")]
    if metric == [1.0] * 6:     # the vector kernel it replaces stays behind GAAST_FLAG_NO_MFMA
        variants += [(ga.F32, ga.FLAG_NO_MFMA, "product_dense[", "k_gp_dense<float,"), (ga.F64, ga.FLAG_NO_MFMA, "product_dense[", "k_gp_dense<double,")]
    # (a general metric's list has general coefficients: the reference's summation order stays available on k_product_csr)
    _check(6, metric, None, variants, batch=1003, seed=600 + int(sum(metric)), eps_factor=8 if general else 4, exact_order_too=general)


@pytest.mark.parametrize("dtype", [ga.F32, ga.F64])
def test_n6_partial_operands_projected_result_accumulation_and_shared_rows(dtype):
    """k_gp_mfma6 beyond the plain product: operands that hold only some grades (components nobody loads stay zero in the
    images), a grade-projected result (stores skipped), a sum of two products into one buffer (beta = 1) and one operand row
    shared by every item (stride 0)."""
    n, batch = 6, 77
    rng = np.random.default_rng(66)
    npdt, eps = (np.float32, 2.0 ** -23) if dtype == ga.F32 else (np.float64, 2.0 ** -52)
    metric = [1.0, 1.0, -1.0, 1.0, 0.0, 1.0]
    lg, rg, og = [0, 1, 2, 3, 4, 6], [1, 2, 3, 4, 5], [0, 2, 3, 5, 6]
    build = lambda B: (B.input(0, lg, n) * B.input(1, rg, n)).gselect(og) + (B.input(2, full_grades(n), n) * B.input(1, rg, n)).gselect(og)
    rows = {0: rows_of(n, lg, batch, rng, np.float32), 1: rows_of(n, rg, 1, rng, np.float32), 2: rows_of(n, full_grades(n), batch, rng, np.float32)}
    rows64 = {s: r.astype(np.float64) for s, r in rows.items()}
    want, wmask = oracle_eval_batch(build, metric, rows64, batch)
    got, mask, spec = hip_eval_batch(build, metric, {s: r.astype(npdt) for s, r in rows.items()}, batch, dtype=dtype)
    assert mask == wmask
    kern = "k_gp_mfma6<float," if dtype == ga.F32 else "k_gp_mfma6<double,"
    assert sum(kern in l for l in spec.launches()) == 2, spec.launches()
    ogl = [k for k in range(n + 1) if (wmask >> k) & 1]
    for i in range(batch):
        S = abs_terms_bound(n, row_to_bits(n, lg, rows64[0][i]), row_to_bits(n, rg, rows64[1][0])) + \
            abs_terms_bound(n, row_to_bits(n, full_grades(n), rows64[2][i]), row_to_bits(n, rg, rows64[1][0]))
        bound = 4 * eps * bits_to_row(n, ogl, S) + 1e-300
        err = np.abs(got[i].astype(np.float64) - want[i])
        assert np.all(err <= bound), (i, float((err / bound).max()))


def test_n7_parity_pure_products_run_in_cl6_on_the_matrix_cores():
    """even x even and odd x even at n = 7 are ONE product in the even subalgebra Cl(6) (plan.cpp rewrite 7): k_gp_mfma6"""
    n = 7
    for lpar, rpar in (("even", "even"), ("odd", "even")):
        grades = {"even": EVEN(n), "odd": ODD(n)}
        _check(n, [1.0, 1.0, -1.0, 1.0, 1.0, -1.0, 1.0], grades[lpar],
               [(ga.F32, 0, "product_dense_mfma[", "k_gp_mfma6<float,false,"), (ga.F64, 0, "product_dense_mfma[", "k_gp_mfma6<double,false,")],
               batch=33, seed=770, right_grades=grades[rpar], label_has=f"{lpar} x {rpar} in Cl(6)")


@pytest.mark.parametrize("n,dtype,metric", [(9, ga.F64, None), (10, ga.F32, None), (12, ga.F64, None), (9, ga.F64, [1.0, -1.0, 2.0, 1.0, 0.5, -1.0, 1.0, -4.0, 1.0])])
def test_the_versor_inverse_beyond_a_fused_slab_is_one_launch(n, dtype, metric):
    """a.vinv() = a.rev() * a.norm_sq().sinv() (expr.rs:363-371), a even, where the rows no longer fit a fused slab: round 3 ran three
    launches (a product into one scalar component -- a single row of 2^(n-1) terms on a thread-per-row kernel --, k_scalar_unary, a
    product of one-term rows).  Round 4: ONE k_reduce_scale launch -- reduction in the reference's order, correctly rounded division,
    scaling -- bit for bit the oracle (f32: its F32 mode) and the unfused plan; also with a general diagonal metric (coefficients
    other than +-1 in both products)."""
    even = EVEN(n)
    build = lambda B: B.input(0, even, n).vinv()
    batch = 41
    npdt = np.float32 if dtype == ga.F32 else np.float64
    rng = np.random.default_rng(300 + n)
    rows = {0: rows_of(n, even, batch, rng, npdt)}
    rows64 = {0: rows[0].astype(np.float64)}
    alg = metric if metric else n
    want, wmask = oracle_eval_batch(build, alg, rows64, batch, mode=ogm.EVAL_RELEASE if dtype == ga.F64 else ogm.EVAL_F32)
    got, mask, spec = hip_eval_batch(build, alg, rows, batch, dtype=dtype)
    assert mask == wmask
    assert len(spec.launches()) == 1 and spec.launches()[0].startswith("reduce_scale[") and "k_reduce_scale<" in spec.launches()[0], spec.launches()
    if metric is None:
        # +-1 coefficients, the reduction a signed sum of squares of the very row that is scaled: without GAAST_FLAG_EXACT_ORDER one
        # wave per item keeps the row in registers (read ONCE), lane-parallel partial sums joined by DPP butterflies.  The terms here
        # are all positive, so the REFERENCE's sequential sum of 2^(n-1) terms is itself up to ~sqrt(terms) eps away from the true
        # norm (measured: 14 eps at n = 12) while 64 short chains + a butterfly stay within 1 eps of it: the two differ by the
        # reference's own rounding error.  Asserted: (a) every component within 6 eps of the TRUE value +-a_k / N (N summed
        # exactly, math.fsum) -- closer than the reference is --, (b) within 32 eps of the reference's, (c) with the flag, the
        # reference's bits (sixteen items per wave, sequential chains).
        import math
        assert "k_reduce_scale_wave<" in spec.launches()[0], spec.launches()
        eps = float(np.finfo(npdt).eps)
        a64 = rows64[0]
        n_true = np.array([math.fsum(r * r) for r in a64])
        true = np.sign(want) * np.abs(a64) / n_true[:, None]
        g64 = got.astype(np.float64)
        assert np.all(np.abs(g64 - true) <= 6 * eps * np.abs(true) + 1e-300), float((np.abs(g64 - true) / (eps * np.abs(true) + 1e-300)).max())
        assert np.all(np.abs(g64 - want) <= 32 * eps * np.abs(want) + 1e-300), float((np.abs(g64 - want) / (eps * np.abs(want) + 1e-300)).max())
        assert not np.array_equal(got.astype(np.float64), want)                 # (the re-ordered sum really ran)
        got, mask, spec = hip_eval_batch(build, alg, rows, batch, dtype=dtype, flags=ga.FLAG_EXACT_ORDER)
        assert len(spec.launches()) == 1 and "k_reduce_scale<" in spec.launches()[0] and "k_reduce_scale_wave" not in spec.launches()[0], spec.launches()
    assert np.array_equal(got.astype(np.float64), want)
    three, _, spec3 = hip_eval_batch(build, alg, rows, batch, dtype=dtype, flags=ga.FLAG_DEBUG_NO_CHAIN)
    assert len(spec3.launches()) == 3 and np.array_equal(three.astype(np.float64), want), spec3.launches()
    assert any("k_scalar_unary" in l or "scalar_inversion" in l for l in spec3.launches()), spec3.launches()


@pytest.mark.parametrize("name,dtype", [("vinv8", ga.F64), ("vinv8", ga.F32), ("proj12", ga.F64), ("proj12", ga.F32), ("unary12", ga.F64)])
def test_medium_programs_run_as_straight_line_code_over_slabs_in_lds(name, dtype):
    """Programs whose slab is beyond the registers of the specialised kernel (160 / 200 elements) but short (<= 2048 comp-muls): the
    interpreter's schedule as hiprtc-compiled straight-line code over slabs that stay in LDS (round 4; the LDS interpreter remains the
    fallback).  The versor inverse at n = 8 (slab 259), the projection KAT of eval.rs:152-163 at n = 12 (slab 171), and a chain of
    unary arms (Negation, Reverse, GradeInvolution, Addition) over bivector rows at n = 12: bit for bit the oracle and the interpreter."""
    npdt = np.float32 if dtype == ga.F32 else np.float64
    rng = np.random.default_rng(77)
    batch = 131
    if name == "vinv8":
        n = 8
        build = lambda B: B.input(0, EVEN(n), n).vinv()
        rows = {0: rows_of(n, EVEN(n), batch, rng, npdt)}
    elif name == "proj12":
        n = 12
        build = lambda B: (lambda v, bv: (v & bv) & bv.vinv())(B.input(0, [1], n), B.input(1, [2], n))
        rows = {0: rows_of(n, [1], batch, rng, npdt), 1: rows_of(n, [2], batch, rng, npdt)}
    else:
        n = 12
        build = lambda B: (-(B.input(0, [2], n).rev()) + B.input(1, [2], n).ginvol()).rev() * B.input(2, [0], n)
        rows = {0: rows_of(n, [2], batch, rng, npdt), 1: rows_of(n, [2], batch, rng, npdt), 2: rows_of(n, [0], batch, rng, npdt)}
    rows64 = {s_: r.astype(np.float64) for s_, r in rows.items()}
    want, wmask = oracle_eval_batch(build, n, rows64, batch, mode=ogm.EVAL_RELEASE if dtype == ga.F64 else ogm.EVAL_F32)
    got, mask, spec = hip_eval_batch(build, n, rows, batch, dtype=dtype)
    assert mask == wmask
    assert len(spec.launches()) == 1 and spec.launches()[0].startswith("ast_jit["), spec.launches()
    # where the slab lives is decided by a TRIAL compilation (round 4): up to 256 / 320 elements one item per thread is tried first and
    # kept when the compiled kernel leaves two waves per SIMD -- the projection at n = 12 (slab 171: 222 registers; 0.75 of the HBM
    # roof against 0.44 with its slabs in LDS); the versor inverse at n = 8 (slab 259: the whole row is live until it is scaled, 310
    # registers) falls back to slabs in LDS, as does anything bigger
    in_lds = "slab in LDS" in spec.launches()[0]
    if name == "proj12" or (name, dtype) == ("vinv8", ga.F64):
        assert in_lds == (name != "proj12"), spec.launches()
    assert np.array_equal(got.astype(np.float64), want)
    interp, _, spec2 = hip_eval_batch(build, n, rows, batch, dtype=dtype, flags=ga.FLAG_NO_JIT)
    assert len(spec2.launches()) == 1 and spec2.launches()[0].startswith("ast_fused[") and np.array_equal(interp.astype(np.float64), want), spec2.launches()


@pytest.mark.parametrize("n,dtype", [(8, ga.F64), (8, ga.F32), (9, ga.F64), (12, ga.F64)])
def test_a_list_with_few_long_rows_runs_on_the_specialised_kernel_with_the_copy_folded_in(n, dtype):
    """d = (a + b * c).g(2) (README.md:20-22, BASELINE configs[0]) beyond R^3: b * c projected on grade 2 is a list of C(n,2) rows
    of 2^n terms -- too sparse for the dense kernels, too few rows for k_product_ell's thread per row.  Round 4: the list runs on
    the specialised chain kernel alone (lane = (row, item)), the covering copy of a's grade 2 folded into its accumulators: ONE
    launch.  With GAAST_FLAG_EXACT_ORDER bit for bit the oracle (f32: its F32 mode) and the plans without run-time compilation /
    without fusion; by default long rows may be summed in slices (n = 12: 4,096-term rows): within 4 eps sum |terms|."""
    full = full_grades(n)
    build = lambda B: (B.input(0, full, n) + B.input(1, full, n) * B.input(2, full, n)).g(2)
    batch = 77 if n < 12 else 5
    npdt = np.float32 if dtype == ga.F32 else np.float64
    rng = np.random.default_rng(120 + n)
    rows = {s_: rows_of(n, full, batch, rng, npdt) for s_ in range(3)}
    rows64 = {s_: r.astype(np.float64) for s_, r in rows.items()}
    want, wmask = oracle_eval_batch(build, n, rows64, batch, mode=ogm.EVAL_RELEASE if dtype == ga.F64 else ogm.EVAL_F32)
    got, mask, spec = hip_eval_batch(build, n, rows, batch, dtype=dtype, flags=ga.FLAG_EXACT_ORDER)
    assert mask == wmask == 0x4
    assert len(spec.launches()) == 1 and "copy_grades_from" in spec.launches()[0] and "gaast_chain<" in spec.launches()[0] and "one list" in spec.launches()[0], spec.launches()
    assert "re-ordered" not in spec.launches()[0]
    assert np.array_equal(got.astype(np.float64), want)
    two, _, spec2 = hip_eval_batch(build, n, rows, batch, dtype=dtype, flags=ga.FLAG_NO_JIT | ga.FLAG_EXACT_ORDER)
    assert len(spec2.launches()) == 2 and "copy_grades_from" in spec2.launches()[0] and "k_product_ell<" in spec2.launches()[1], spec2.launches()
    assert np.array_equal(two.astype(np.float64), want)
    # the unfused plan (one kernel per eval.rs arm, every operand materialised): zero fills, add_grades_from copies, the product
    three, _, spec3 = hip_eval_batch(build, n, rows, batch, dtype=dtype, flags=ga.FLAG_NO_FUSION)
    assert len(spec3.launches()) == 7 and sum("add_grades_from" in l for l in spec3.launches()) == 3, spec3.launches()
    assert np.array_equal(three.astype(np.float64), want)
    # default flags: tolerance mode (slices of long rows where lanes are free)
    tol, tmask, spec4 = hip_eval_batch(build, n, rows, batch, dtype=dtype)
    assert tmask == wmask and len(spec4.launches()) == 1
    eps = 2.0 ** -23 if dtype == ga.F32 else 2.0 ** -52
    for i in range(batch):
        S = abs_terms_bound(n, row_to_bits(n, full, rows64[1][i]), row_to_bits(n, full, rows64[2][i]))
        bound = 4 * eps * (bits_to_row(n, [2], S) + np.abs(want[i])) + 1e-300
        err = np.abs(tol[i].astype(np.float64) - want[i])
        assert np.all(err <= bound), (i, float((err / bound).max()))


@pytest.mark.parametrize("dtype", [ga.F64, ga.F32])
@pytest.mark.parametrize("shape", ["scaled", "plain", "mixed_grades"])
def test_runs_of_element_wise_arms_are_one_pass(shape, dtype):
    """GradedObj, Negation, Reverse, GradeInvolution, Addition (eval.rs:45-60, 87-102) on rows too big for a fused slab: round 3 ran
    one launch per arm (k_axpy_map, k_flip, ... each a full read-modify-write of the buffer).  Round 4: the run is ONE k_elementwise
    pass, every component executing its own statements in program order -- with the product that scales the result by a scalar
    operand as its epilogue when there is one.  Bit for bit the oracle (f32: its F32 mode) and the one-launch-per-arm plan;
    Q1 of SURVEY section 7 included (the unary arms act on the whole accumulator)."""
    n, batch = 12, 23
    npdt = np.float32 if dtype == ga.F32 else np.float64
    rng = np.random.default_rng(5150)
    if shape == "scaled":
        build = lambda B: (-(B.input(0, [6], n).rev()) + B.input(1, [6], n).ginvol()).rev() * B.input(2, [0], n)
        rows = {0: rows_of(n, [6], batch, rng, npdt), 1: rows_of(n, [6], batch, rng, npdt), 2: rows_of(n, [0], batch, rng, npdt)}
        expect = 1
    elif shape == "plain":
        build = lambda B: -((B.input(0, [5], n) + B.input(1, [5], n).rev()).ginvol() + B.input(0, [5], n))
        rows = {0: rows_of(n, [5], batch, rng, npdt), 1: rows_of(n, [5], batch, rng, npdt)}
        expect = 1
    else:   # operands with several grades each: flips that touch only some of the components (grade 5, 6, 7: Reverse negates 6 and 7)
        build = lambda B: (B.input(0, [5, 6, 7], n).rev() - B.input(1, [5, 6, 7], n)).ginvol()
        rows = {0: rows_of(n, [5, 6, 7], batch, rng, npdt), 1: rows_of(n, [5, 6, 7], batch, rng, npdt)}
        expect = 1
    rows64 = {s_: r.astype(np.float64) for s_, r in rows.items()}
    want, wmask = oracle_eval_batch(build, n, rows64, batch, mode=ogm.EVAL_RELEASE if dtype == ga.F64 else ogm.EVAL_F32)
    got, mask, spec = hip_eval_batch(build, n, rows, batch, dtype=dtype)
    assert mask == wmask
    assert len(spec.launches()) == expect and spec.launches()[0].startswith("elementwise[") and "k_elementwise<" in spec.launches()[0], spec.launches()
    assert np.array_equal(got.view(np.uint32 if dtype == ga.F32 else np.uint64), want.astype(npdt).view(np.uint32 if dtype == ga.F32 else np.uint64))   # signs of zeros too
    per_arm, _, spec2 = hip_eval_batch(build, n, rows, batch, dtype=dtype, flags=ga.FLAG_DEBUG_NO_CHAIN)
    assert len(spec2.launches()) > 2 and any("k_flip<" in l for l in spec2.launches()) and any("k_axpy_map<" in l for l in spec2.launches()), spec2.launches()
    assert np.array_equal(per_arm.view(np.uint32 if dtype == ga.F32 else np.uint64), got.view(np.uint32 if dtype == ga.F32 else np.uint64))
