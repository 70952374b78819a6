"""BASELINE.json's full sizes on the GPU, checked through size-independent properties.

The oracle cannot run 65,536 products of 16.7 M terms, so at full size the checks are identities of the
algebra that hold EXACTLY in floating point (every term is a component times 0 or +-1, so no rounding
happens) and therefore compare every component of every item bit for bit, plus sampled items against
the independent float64 bitmask convolution:

  * 1 * B = B and I * B = the signed permutation of B by the pseudoscalar (algebra.rs:73-83)
    (the opt-in matrix path transforms B back and forth, so there the same identities are checked
    to its norm-wise bound);
  * one-hot blades: e_a * e_b = coeff * e_{a^b};
  * v * v = |v|^2: grades 1 and >= 3 exactly zero, grade 2 zero to one rounding;
  * R = 1  =>  R X ~R = X (config 5).

Inputs live in torch tensors on the device and are wrapped (gaast_hip_mv_wrap), as in bench.py.
"""
import numpy as np
import pytest

import gaast_amd as ga
from helpers import bits_to_row, blades_in_row_order, full_grades, gp_bits, n_choose_k, row_to_bits

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

FULL = {12: 65536, 8: 1 << 20}            # BASELINE configs[2], configs[1]
PATHS = [(0, "product_dense"), (ga.FLAG_SPINOR_GEMM, "product_spinor_gemm")]


def _spec(n, flags, metric=None):
    a, b = ga.mv(ga.Input(0, full_grades(n), n)), ga.mv(ga.Input(1, full_grades(n), n))
    return (a * b).specialize(ga.MetricAlgebra(metric or [1.0] * n), dtype=ga.F32, flags=flags)


def _run(spec, n, ta, tb):
    batch = max(ta.shape[0], tb.shape[0])
    out_t = torch.empty((batch, 1 << n), device="cuda", dtype=torch.float32)
    ins = [ga.DeviceMV.wrap_tensor(ta, n, full_grades(n)), ga.DeviceMV.wrap_tensor(tb, n, full_grades(n))]
    out = ga.DeviceMV.wrap_tensor(out_t, n, full_grades(n))
    spec.eval_batch(ins, batch, out=out)
    torch.cuda.synchronize()
    return out_t


@pytest.mark.parametrize("n", [12, 8])
@pytest.mark.parametrize("flags,kernel", PATHS)
def test_unit_and_pseudoscalar_times_b_are_exact_at_full_batch(n, flags, kernel):
    batch, N = FULL[n], 1 << n
    spec = _spec(n, flags)
    assert any(kernel in l for l in spec.launches()), spec.launches()
    gen = torch.Generator(device="cuda")
    gen.manual_seed(n)
    tb = torch.rand((batch, N), generator=gen, device="cuda", dtype=torch.float32) * 2 - 1
    blades = blades_in_row_order(n, full_grades(n))
    pos_of = np.zeros(N, dtype=np.int64)
    pos_of[blades] = np.arange(N)
    # left operand shared by every item (batch 1): the scalar 1, then the pseudoscalar
    one = torch.zeros((1, N), device="cuda", dtype=torch.float32)
    one[0, 0] = 1.0
    def same(got, want):
        if not flags:
            return torch.equal(got, want)                # every term is b * (0 or +-1): no rounding anywhere
        # matrix path: B goes through two transforms, so the identity holds to the path's norm-wise bound
        bound = 64 * 2.0 ** -23 * tb.double().norm(dim=1, keepdim=True)
        return bool(torch.all((got.double() - want.double()).abs() <= bound))

    assert same(_run(spec, n, one, tb), tb)
    ps = torch.zeros((1, N), device="cuda", dtype=torch.float32)
    ps[0, N - 1] = 1.0
    got = _run(spec, n, ps, tb)
    alg = ga.MetricAlgebra([1.0] * n)
    src = np.zeros(N, dtype=np.int64)
    sign = np.zeros(N, dtype=np.float32)
    for b in range(N):                                   # I * e_b = coeff * e_{I^b}
        res, coeff = alg.ortho_basis_blades_gp(N - 1, b)
        src[pos_of[res]] = pos_of[b]
        sign[pos_of[res]] = coeff
    want = tb[:, torch.from_numpy(src).cuda()] * torch.from_numpy(sign).cuda()
    assert same(got, want)


@pytest.mark.parametrize("flags,kernel", PATHS)
def test_r12_full_batch_sampled_items_against_bitmask_convolution(flags, kernel):
    n, batch, N = 12, FULL[12], 4096
    spec = _spec(n, flags)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(99)
    ta = torch.rand((batch, N), generator=gen, device="cuda", dtype=torch.float32) * 2 - 1
    tb = torch.rand((batch, N), generator=gen, device="cuda", dtype=torch.float32) * 2 - 1
    got = _run(spec, n, ta, tb)
    for i in (0, 1, 255, 256, 32767, 40000, batch - 2, batch - 1):      # first / last / workgroup boundaries
        ra, rb = ta[i].cpu().numpy().astype(np.float64), tb[i].cpu().numpy().astype(np.float64)
        want = bits_to_row(n, full_grades(n), gp_bits(n, [1.0] * n, row_to_bits(n, full_grades(n), ra),
                                                      row_to_bits(n, full_grades(n), rb)))
        err = np.abs(got[i].cpu().numpy().astype(np.float64) - want).max()
        if flags:
            assert err <= 64 * 2.0 ** -23 * np.linalg.norm(ra) * np.linalg.norm(rb), (i, err)
        else:
            absum = bits_to_row(n, full_grades(n), gp_bits(n, [1.0] * n, row_to_bits(n, full_grades(n), ra),
                                                           row_to_bits(n, full_grades(n), rb), absolute=True))
            assert np.all(np.abs(got[i].cpu().numpy().astype(np.float64) - want) <= 4 * 2.0 ** -23 * absum), i


@pytest.mark.parametrize("flags,kernel", PATHS)
def test_r12_vector_squared_is_scalar_at_full_batch(flags, kernel):
    """v v = |v|^2.  Grades 1 and >= 3 only ever receive 0 * x terms: exactly zero on every path.  Grade 2
    is a_i a_j - a_j a_i: zero up to one rounding of the product (the dense kernels accumulate with FMA,
    so the second term does not cancel the rounded first one exactly -- the reference, without FMA, does)."""
    n, batch, N = 12, FULL[12], 4096
    spec = _spec(n, flags)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(5)
    tv = torch.zeros((batch, N), device="cuda", dtype=torch.float32)
    tv[:, 1:13] = torch.rand((batch, 12), generator=gen, device="cuda", dtype=torch.float32) * 2 - 1
    got = _run(spec, n, tv, tv)
    want = (tv[:, 1:13].double() ** 2).sum(dim=1)
    if flags:
        bound = 64 * 2.0 ** -23 * want.unsqueeze(1)                      # |v|_2 |v|_2
        ref = torch.zeros_like(got, dtype=torch.float64)
        ref[:, 0] = want
        assert torch.all((got.double() - ref).abs() <= bound)
        return
    assert torch.count_nonzero(got[:, 1:13]).item() == 0 and torch.count_nonzero(got[:, 79:]).item() == 0
    assert torch.all(got[:, 13:79].abs().double() <= 2.0 ** -23 * want.unsqueeze(1))
    assert torch.all((got[:, 0].double() - want).abs() <= 12 * 2.0 ** -23 * want)


def test_cl41_identity_rotor_sandwich_at_full_batch():
    """config 5 at 4 M items, f64: R = 1 gives R X ~R = X bit for bit; sampled random items vs the oracle."""
    from helpers import oracle_eval_batch
    batch, n = 1 << 22, 5
    metric = [1.0, 1.0, 1.0, 1.0, -1.0]
    build = lambda B: B.input(0, [0, 2, 4], n) * B.input(1, [1], n) * B.input(0, [0, 2, 4], n).rev()
    r, x = ga.mv(ga.Input(0, [0, 2, 4], n)), ga.mv(ga.Input(1, [1], n))
    spec = (r * x * r.rev()).specialize(ga.MetricAlgebra(metric))
    gen = torch.Generator(device="cuda")
    gen.manual_seed(41)
    tx = torch.rand((batch, 5), generator=gen, device="cuda", dtype=torch.float64) * 2 - 1
    tr = torch.zeros((batch, 16), device="cuda", dtype=torch.float64)
    tr[:, 0] = 1.0
    out_mask, out_len = spec.output_info()
    out_t = torch.empty((batch, out_len), device="cuda", dtype=torch.float64)
    out = ga.DeviceMV.wrap_tensor(out_t, n, ga.GradeSet(out_mask))
    ins = [ga.DeviceMV.wrap_tensor(tr, n, [0, 2, 4]), ga.DeviceMV.wrap_tensor(tx, n, [1])]
    spec.eval_batch(ins, batch, out=out)
    torch.cuda.synchronize()
    assert out_len == 16                                     # grades {1,3,5}
    assert torch.equal(out_t[:, :5], tx) and torch.count_nonzero(out_t[:, 5:]).item() == 0
    # random rotors: a sample of items, bit-exact against the oracle
    tr2 = torch.rand((batch, 16), generator=gen, device="cuda", dtype=torch.float64) * 2 - 1
    ins = [ga.DeviceMV.wrap_tensor(tr2, n, [0, 2, 4]), ga.DeviceMV.wrap_tensor(tx, n, [1])]
    spec.eval_batch(ins, batch, out=out)
    torch.cuda.synchronize()
    idx = [0, 1, 63, 64, 4095, 1 << 20, batch - 65, batch - 1]
    rows = {0: tr2[idx].cpu().numpy(), 1: tx[idx].cpu().numpy()}
    want, omask = oracle_eval_batch(build, metric, rows, len(idx))
    assert omask == out_mask
    assert np.array_equal(out_t[idx].cpu().numpy(), want)


@pytest.mark.parametrize("n", [12, 8])
def test_scaling_and_sign_symmetries_hold_bit_for_bit_at_full_batch(n):
    """Properties the dense kernels must keep EXACTLY whatever their summation order, at the BASELINE batch, every
    component of every item: (2^k A) B = 2^k (A B) (scaling by a power of two commutes with every rounding),
    (-A) B = -(A B) = A (-B) (rounding is symmetric), and the same for a different power on the right operand."""
    batch, N = FULL[n], 1 << n
    spec = _spec(n, 0)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(77 + n)
    ta = torch.rand((batch, N), generator=gen, device="cuda", dtype=torch.float32) * 2 - 1
    tb = torch.rand((batch, N), generator=gen, device="cuda", dtype=torch.float32) * 2 - 1
    base = _run(spec, n, ta, tb).clone()
    assert torch.equal(_run(spec, n, ta * 8.0, tb), base * 8.0)
    assert torch.equal(_run(spec, n, ta, tb * 0.25), base * 0.25)
    assert torch.equal(_run(spec, n, -ta, tb), -base)
    assert torch.equal(_run(spec, n, ta, -tb), -base)
    # ... and a checksum over the whole batch that changes if any item is mixed up with another: item i scaled by 2^(i % 5)
    scale = (2.0 ** (torch.arange(batch, device="cuda") % 5)).to(torch.float32).unsqueeze(1)
    assert torch.equal(_run(spec, n, ta * scale, tb), base * scale)


def test_linearity_in_the_left_operand_at_full_batch():
    """(A + A') B = A B + A' B to the dense tolerance, every component of every item of the R^12 batch; the bound is
    4 eps of the sum of |terms| bounded from above by |A|_1-type norms, here simply checked against the products'
    own magnitudes: |(A + A') B - A B - A' B| <= 16 eps * 4096 * max|A| max|B|."""
    n, batch, N = 12, FULL[12], 4096
    spec = _spec(n, 0)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(99)
    ta = torch.rand((batch, N), generator=gen, device="cuda", dtype=torch.float32) * 2 - 1
    ta2 = torch.rand((batch, N), generator=gen, device="cuda", dtype=torch.float32) * 2 - 1
    tb = torch.rand((batch, N), generator=gen, device="cuda", dtype=torch.float32) * 2 - 1
    lhs = _run(spec, n, ta + ta2, tb).clone()
    rhs = _run(spec, n, ta, tb).clone()
    rhs += _run(spec, n, ta2, tb)
    err = (lhs.double() - rhs.double()).abs().max().item()
    assert err <= 16 * 2.0 ** -23 * 4096 * 2.0, err
