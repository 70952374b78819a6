"""Fuzz on the GPU: random programs over the reference's operator surface, evaluated by the oracle and by
the HIP evaluator through the C ABI under three plans -- hiprtc-specialised kernel, LDS interpreter kernel,
one launch per eval.rs arm -- must agree BIT FOR BIT (values, grade set of the result, sign of zeros; NaNs
in the same places).  Programs on which the reference panics must fail in the product too."""
import numpy as np
import pytest

import gaast_amd as ga
from fuzz import random_big_program, random_program, realise
from helpers import HipBackend, OracleBackend
from oracle import pyoracle as og

pytestmark = pytest.mark.gpu
N_PROGRAMS = 480
PLANS = [("jit", ga.FLAG_EXACT_ORDER), ("interpreter", ga.FLAG_EXACT_ORDER | ga.FLAG_NO_JIT), ("unfused", ga.FLAG_NO_FUSION)]


def _same_bits(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    if a.shape != b.shape:
        return False
    nan = np.isnan(a)
    if not np.array_equal(nan, np.isnan(b)):
        return False
    return np.array_equal(a[~nan].view(np.uint64), b[~nan].view(np.uint64))


@pytest.mark.parametrize("chunk", range(8))
def test_random_programs_evaluate_bit_exactly(chunk):
    checked = 0
    for seed in range(1000 + chunk * (N_PROGRAMS // 8), 1000 + (chunk + 1) * (N_PROGRAMS // 8)):
        n, metric, spec = random_program(seed)
        try:
            want = realise(spec, OracleBackend(), n).specialize(og.as_algebra(metric)).eval().to_dict()
            panic = None
        except og.OraclePanic as e:
            want, panic = None, e
        for plan, flags in PLANS:
            try:
                got = realise(spec, HipBackend(), n).specialize(metric, flags=flags).eval().to_dict()
                err = None
            except (ga.GaastError, RuntimeError) as e:
                got, err = None, e
            where = f"seed {seed} plan {plan}\n{spec}"
            if panic is not None or err is not None:
                assert panic is not None and err is not None, f"oracle {panic!r} vs product {err!r}: {where}"
                continue
            assert set(got) == set(want), f"grades {sorted(got)} vs {sorted(want)}: {where}"
            for k in want:
                assert _same_bits(got[k], want[k]), f"grade {k}: {got[k]} vs {want[k]}: {where}"
            checked += 1
    assert checked > 0


@pytest.mark.parametrize("chunk", range(4))
def test_random_programs_over_bound_inputs_batched(chunk):
    """the same generator with multivector leaves as bound inputs (re-bound per item) and a batch of 7"""
    from helpers import hip_eval_batch, oracle_eval_batch
    batch, checked = 7, 0
    for seed in range(5000 + chunk * 30, 5000 + (chunk + 1) * 30):
        n, metric, spec = random_program(seed)
        rows = {}
        build = lambda B: realise(spec, B, n, rows={} if B.name == "oracle" else {}, batch=batch)
        # rows are a function of the spec only: collect them once
        realise(spec, OracleBackend(), n, rows=rows, batch=batch)
        if not rows:
            continue
        try:
            want, omask = oracle_eval_batch(build, metric, rows, batch)
            panic = None
        except og.OraclePanic as e:
            want, panic = None, e
        for plan, flags in PLANS:
            try:
                got, hmask, _ = hip_eval_batch(build, metric, rows, batch, flags=flags)
                err = None
            except (ga.GaastError, RuntimeError) as e:
                got, err = None, e
            where = f"seed {seed} plan {plan}\n{spec}"
            if panic is not None or err is not None:
                assert panic is not None and err is not None, f"oracle {panic!r} vs product {err!r}: {where}"
                continue
            assert hmask == omask, where
            assert _same_bits(got, want), where
            checked += 1
    assert checked > 0


def test_random_big_programs_in_reference_order():
    """n = 7, 8 with dense leaves: plans of several launches (k_product_ell / k_product_csr, copies, sign flips);
    GAAST_FLAG_EXACT_ORDER keeps every product in the reference's summation order: bit-exact."""
    kernels = set()
    for seed in range(9000, 9024):
        n, metric, spec = random_big_program(seed)
        want = realise(spec, OracleBackend(), n).specialize(og.as_algebra(metric)).eval().to_dict()
        hs = realise(spec, HipBackend(), n).specialize(metric, flags=ga.FLAG_EXACT_ORDER)
        got = hs.eval().to_dict()
        kernels.update(l.split("[")[0] for l in hs.launches())
        assert set(got) == set(want), f"seed {seed}\n{spec}"
        for k in want:
            assert _same_bits(got[k], want[k]), f"seed {seed} grade {k}\n{spec}"
    assert {"product_ell", "product_csr"} <= kernels, kernels


def test_random_big_programs_default_and_matrix_paths_stay_close():
    """the same big programs on the default plan (dense re-ordered kernels where they apply) in f64 and with the
    opt-in matrix-representation kernels in f32: not bit-exact by design -- checked against the oracle relative
    to the size of the terms (sum of |components| of the result is a cheap stand-in for a full error bound)."""
    dense = spinor = rescaled = 0
    for seed in range(9000, 9024):
        n, metric, spec = random_big_program(seed)
        want = realise(spec, OracleBackend(), n).specialize(og.as_algebra(metric)).eval().to_dict()
        scale = max(1.0, max((float(np.abs(v).max()) for v in want.values() if len(v)), default=1.0))
        hs = realise(spec, HipBackend(), n).specialize(metric)
        got = hs.eval().to_dict()
        dense += any("product_dense" in l for l in hs.launches())
        rescaled += any("rescaled basis" in l for l in hs.launches())
        assert set(got) == set(want)
        for k in want:
            assert np.allclose(got[k], want[k], rtol=0, atol=1e-11 * scale * 4 ** n / 256), f"seed {seed} grade {k}\n{spec}"
        hs = realise(spec, HipBackend(), n).specialize(metric, dtype=ga.F32, flags=ga.FLAG_SPINOR_GEMM)
        got = hs.eval().to_dict()
        spinor += any("product_spinor_gemm" in l for l in hs.launches())
        for k in want:
            assert np.allclose(got[k], want[k], rtol=0, atol=2e-4 * scale), f"seed {seed} grade {k} (f32)\n{spec}"
    assert dense > 0 and spinor > 0 and rescaled > 0, (dense, spinor, rescaled)   # general diagonal metrics reach the dense kernels too
