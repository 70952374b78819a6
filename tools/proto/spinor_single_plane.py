"""Prototype (numpy): the matrix-representation product with ONE real plane per operand.

e_S = i^k X^x Z^z with k = 2u + f.  The parity f is linear: f = alpha.x ^ lambda.z.  After a change of
basis of the 6-bit index spaces (z' = L z, x' = L^-T x, c' = L^-T c, so that c.z = c'.z') lambda'
is a unit vector e_b and alpha' is 0 or e_t (top bit).  Then with W[x][z] = (-1)^u A_S, What = WHT_z(W):

    M_A[c^x][c] = E(p, sigma q),  p = What[x][c], q = What[x][c ^ e_b], sigma = (-1)^(alpha'.x),
    E(p, q) = ((p+q) + i (p-q)) / 2

and E(p,q') E(r,s') = (p s' + q' r)/2 + i (p r - q' s')/2: three real products X = p r, Y = q' s',
Z = (p+q')(r+s').  sigma_A sigma_B contains (-1)^(alpha'.k) only in the real part: the k loop is split
in the halves alpha'.k = 0 / 1 and the real part of the second half is subtracted.
Output: component S(x,z) = (-1)^u Re(i^-f V[x][z]) picks the real or the imaginary plane per (row, z_b):
thread (x, hb) folds ONE plane over bit b and transforms the remaining bits.

Validated here for every signature of n = 6 (8 x 8 matrices) and a few of n = 8 against gp_bits.
"""
import itertools
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from spinor_rep import pauli_tables  # noqa: E402


def parity(v):
    return bin(int(v)).count("1") & 1


def wht(a):
    a = a.copy()
    n = a.shape[-1]
    h = 1
    while h < n:
        for i in range(n):
            if not i & h:
                x, y = a[..., i].copy(), a[..., i | h].copy()
                a[..., i], a[..., i | h] = x + y, x - y
        h <<= 1
    return a


def solve_linear_parity(m, X, Z, K):
    """alpha, lambda with (K & 1) == alpha.x ^ lambda.z for every blade"""
    alpha = lam = 0
    f = {(int(x), int(z)): int(k) & 1 for x, z, k in zip(X, Z, K)}
    assert f[(0, 0)] == 0
    for j in range(m):
        alpha |= f[(1 << j, 0)] << j
        lam |= f[(0, 1 << j)] << j
    for (x, z), v in f.items():
        assert v == parity(x & alpha) ^ parity(z & lam)
    return alpha, lam


def gf2_inv(rows, m):
    """inverse of the m x m GF(2) matrix given as row bitmasks; returns row bitmasks"""
    a = [(rows[i], 1 << i) for i in range(m)]
    for col in range(m):
        piv = next(i for i in range(col, m) if (a[i][0] >> col) & 1)
        a[col], a[piv] = a[piv], a[col]
        for i in range(m):
            if i != col and (a[i][0] >> col) & 1:
                a[i] = (a[i][0] ^ a[col][0], a[i][1] ^ a[col][1])
    return [a[i][1] for i in range(m)]


def mat_apply(rows, v):
    return sum(parity(r & v) << i for i, r in enumerate(rows))


def transpose(rows, m):
    return [sum(((rows[i] >> j) & 1) << i for i in range(m)) for j in range(m)]


def choose_basis(m, alpha, lam):
    """rows of L with row[b] = lambda (b = m-1, or m-2 when lambda.alpha = 0 and both nonzero) and
    L alpha in {0, e_(m-1)}.  Returns (rows, lam_bit or -1, alpha' nonzero?)."""
    top = m - 1
    rows = [None] * m
    if lam == 0:
        lam_bit = -1
        if alpha:
            want_one = [top]
        else:
            want_one = []
    else:
        if alpha and parity(lam & alpha) == 0:
            lam_bit = m - 2      # lambda on the second bit so that alpha' can sit on the top one
            want_one = [top]
        else:
            lam_bit = top
            want_one = [top] if alpha else []
        rows[lam_bit] = lam
    # fill the remaining rows: r.alpha = 1 for the row in want_one (unless it is the lambda row), else 0
    span = []

    def independent(v):
        r = v
        for b in span:
            r = min(r, r ^ b)
        return r != 0

    def add(v):
        r = v
        for b in span:
            r = min(r, r ^ b)
        span.append(r)
        span.sort(reverse=True)

    if rows[lam_bit] is not None and lam_bit >= 0:
        add(lam)
    for i in range(m):
        if rows[i] is not None:
            continue
        need = 1 if i in want_one else 0
        for v in range(1, 1 << m):
            if parity(v & alpha) == need and independent(v):
                rows[i] = v
                add(v)
                break
        assert rows[i] is not None
    a2 = mat_apply(rows, alpha)
    assert a2 in (0, 1 << top), (alpha, lam, rows, a2)
    if lam_bit >= 0:
        linv_t = transpose(gf2_inv(rows, m), m)
        assert mat_apply(linv_t, lam) == 1 << lam_bit
    return rows, lam_bit, a2 != 0


def product_single_plane(n, metric, A_bits, B_bits):
    m = n // 2
    D = 1 << m
    X, Z, K = pauli_tables(metric)
    alpha, lam = solve_linear_parity(m, X, Z, K)
    L, lam_bit, has_alpha = choose_basis(m, alpha, lam)
    LinvT = transpose(gf2_inv(L, m), m)
    top = m - 1
    xs = np.array([mat_apply(LinvT, x) for x in X])      # x' = L^-T x
    zs = np.array([mat_apply(L, z) for z in Z])          # z' = L z
    us = (K >> 1) & 1
    fs = K & 1
    for x, z, f in zip(xs, zs, fs):                      # f = x'_top [alpha'] ^ z'_lam_bit
        assert f == ((x >> top) & 1 if has_alpha else 0) ^ ((z >> lam_bit) & 1 if lam_bit >= 0 else 0)
    # one real plane per operand; the right operand is stored shifted by x: sign (-1)^|x&z|
    WA = np.zeros((D, D))
    WB = np.zeros((D, D))
    for S in range(1 << n):
        s = -1.0 if us[S] else 1.0
        WA[xs[S], zs[S]] = s * A_bits[S]
        WB[xs[S], zs[S]] = s * B_bits[S] * (-1.0 if parity(xs[S] & zs[S]) else 1.0)
    HA, HB = wht(WA), wht(WB)
    lam_off = (1 << lam_bit) if lam_bit >= 0 else 0
    Cre = np.zeros((D, D))
    Cim = np.zeros((D, D))
    kappa_bit = top if has_alpha else -1
    for r in range(D):
        for c in range(D):
            rho = -1.0 if has_alpha and (r >> top) & 1 else 1.0
            gam = -1.0 if has_alpha and (c >> top) & 1 else 1.0
            halves = [[0.0, 0.0, 0.0], [0.0, 0.0, 0.0]]
            for k in range(D):
                p, q = HA[r ^ k, k], HA[r ^ k, k ^ lam_off]
                rr, ss = HB[k ^ c, k], HB[k ^ c, k ^ lam_off]
                q, ss = rho * q, gam * ss
                hh = halves[(k >> kappa_bit) & 1 if kappa_bit >= 0 else 0]
                hh[0] += p * rr
                hh[1] += q * ss
                hh[2] += (p + q) * (rr + ss)
            (x0, y0, z0), (x1, y1, z1) = halves
            Cre[r, c] = 0.5 * ((z0 - x0 - y0) - (z1 - x1 - y1))
            Cim[r, c] = 0.5 * ((x0 - y0) + (x1 - y1))
    # C stored skewed by row: S[x][r] = C[r][r ^ x]
    Sre = np.zeros((D, D))
    Sim = np.zeros((D, D))
    for r in range(D):
        for c in range(D):
            Sre[r ^ c, r] = Cre[r, c]
            Sim[r ^ c, r] = Cim[r, c]
    V = np.zeros((D, D))
    for x in range(D):
        xi = (x >> top) & 1 if has_alpha else 0
        for hb in range(2 if lam_bit >= 0 else 1):
            plane = Sim if (xi ^ hb) else Sre
            if lam_bit >= 0:
                idx0 = [j for j in range(D) if not (j >> lam_bit) & 1]
                d = np.array([plane[x, j] + (-1.0 if hb else 1.0) * plane[x, j | lam_off] for j in idx0])
                t = wht(d)                                   # over the remaining m-1 bits, in idx0 order
                for jj, j in enumerate(idx0):
                    V[x, j | (hb << lam_bit)] = t[jj] / D
            else:
                V[x, :] = wht(plane[x, :]) / D
    out = np.zeros(1 << n)
    for S in range(1 << n):
        s = (-1.0 if us[S] else 1.0) * (-1.0 if parity(xs[S] & zs[S]) else 1.0)
        out[S] = s * V[xs[S], zs[S]]
    return out, (alpha, lam, lam_bit, has_alpha)


if __name__ == "__main__":
    from helpers import gp_bits
    rng = np.random.default_rng(1)
    seen = set()
    n = 6
    for signs in itertools.product([1.0, -1.0], repeat=n):
        metric = list(signs)
        A, B = rng.uniform(-1, 1, 1 << n), rng.uniform(-1, 1, 1 << n)
        got, info = product_single_plane(n, metric, A, B)
        err = np.abs(got - gp_bits(n, metric, A, B)).max()
        seen.add(info[2:])
        assert err < 1e-12, (metric, info, err)
    print("n = 6: all 64 signatures OK; (lam_bit, alpha') cases seen:", sorted(seen))
    n = 8
    for metric in ([1.0] * 8, [1.0, -1.0] * 4, [-1.0, 1.0, 1.0, 1.0, -1.0, -1.0, 1.0, -1.0]):
        A, B = rng.uniform(-1, 1, 1 << n), rng.uniform(-1, 1, 1 << n)
        got, info = product_single_plane(n, metric, A, B)
        print(metric, info, "max |err| =", np.abs(got - gp_bits(n, metric, A, B)).max())
