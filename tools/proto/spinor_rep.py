"""Prototype (numpy): the geometric product of Cl(p,q), n = p+q even, through the complex matrix
representation built from Pauli strings (Jordan-Wigner).  Validates the tables the GPU kernel uses."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))

def pauli_tables(metric):
    n = len(metric); m = n // 2
    assert n % 2 == 0 and all(g in (1.0, -1.0) for g in metric)
    gam = []
    for v in range(n):
        j = v // 2
        if v % 2 == 0: x, z, k = 1 << j, (1 << j) - 1, 0
        else:          x, z, k = 1 << j, (1 << (j + 1)) - 1, 1
        if metric[v] < 0: k += 1
        gam.append((x, z, k % 4))
    N = 1 << n
    X = np.zeros(N, np.int64); Z = np.zeros(N, np.int64); K = np.zeros(N, np.int64)
    for S in range(N):
        x = z = k = 0
        for v in range(n):
            if (S >> v) & 1:
                gx, gz, gk = gam[v]
                k = (k + gk + 2 * bin(z & gx).count("1")) % 4
                x ^= gx; z ^= gz
        X[S], Z[S], K[S] = x, z, k
    return X, Z, K

def to_matrix(A_bits, X, Z, K, m):
    D = 1 << m
    M = np.zeros((D, D), complex)
    c = np.arange(D)
    for S, a in enumerate(A_bits):
        if a == 0: continue
        sign = np.where(np.array([bin(int(ci) & int(Z[S])).count("1") for ci in c]) & 1, -1.0, 1.0)
        M[c ^ X[S], c] += a * (1j ** K[S]) * sign
    return M

def from_matrix(M, X, Z, K, m):
    D = 1 << m
    c = np.arange(D)
    out = np.zeros(len(X))
    for S in range(len(X)):
        sign = np.where(np.array([bin(int(ci) & int(Z[S])).count("1") for ci in c]) & 1, -1.0, 1.0)
        v = (M[c ^ X[S], c] * sign).sum() / D
        out[S] = (v * (1j ** (-K[S]))).real
    return out

if __name__ == "__main__":
    from helpers import gp_bits
    rng = np.random.default_rng(0)
    for metric in ([1.0] * 6, [1.0, 1.0, -1.0, 1.0], [1.0, -1.0, -1.0, -1.0, 1.0, -1.0]):
        n = len(metric); m = n // 2
        X, Z, K = pauli_tables(metric)
        assert len(set(zip(X.tolist(), Z.tolist()))) == 1 << n      # bijection blade -> (x, z)
        A, B = rng.uniform(-1, 1, 1 << n), rng.uniform(-1, 1, 1 << n)
        C = from_matrix(to_matrix(A, X, Z, K, m) @ to_matrix(B, X, Z, K, m), X, Z, K, m)
        ref = gp_bits(n, metric, A, B)
        print(metric, "max |err| =", np.abs(C - ref).max())
