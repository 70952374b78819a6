"""CPU emulation of k_gp_mfma6's tile decomposition (round 4), checked against the bitmask form of the geometric product
C[a ^ b] += s(a, b) m(a & b) A[a] B[b] (algebra.rs:73-83, 199-209) for every +-1 / 0 diagonal metric pattern class.

blade = (top2 | hi2 | lo2).  16x16 tile: row = (u = a_top, x = c_lo), column = (v = b_top, y = c_hi); K = (a_hi, b_lo) =
(step s, kq).  Tile element T[(u,x),(v,y)] = sum_{s,kq} Aop * Bop; component (w, y, x) = sum_u T[(u,x),(u^w,y)].
Signs factor into (row, k) [A image], (k, column) [B image], (row, column) [result]."""
import itertools
import numpy as np


def rp(a, b):  # reorder parity within a group: #{(p, q): p in a, q in b, p > q} mod 2
    c = 0
    for q in range(8):
        if (b >> q) & 1:
            c += bin(a >> (q + 1)).count("1")
    return c & 1


def pc(x):
    return bin(x).count("1")


def direct(A, B, neg, zero, n=6):
    N = 1 << n
    C = np.zeros(N)
    for a in range(N):
        for b in range(N):
            if a & b & zero:
                continue
            s = rp(a, b) ^ (pc(a & b & neg) & 1)
            C[a ^ b] += (-1.0 if s else 1.0) * A[a] * B[b]
    return C


def tiled(A, B, neg, zero):
    NEG_L, NEG_H, NEG_T = neg & 3, (neg >> 2) & 3, (neg >> 4) & 3
    Z_L, Z_H, Z_T = zero & 3, (zero >> 2) & 3, (zero >> 4) & 3
    # operand images in MFMA order: Aop[(u,x)][k=(s,kq)], Bop[k][(v,y)]
    Aop = np.zeros((4, 4, 4, 4))  # u, x, s, kq
    Bop = np.zeros((4, 4, 4, 4))  # s, kq, v, y
    for a in range(64):
        u, ah, al = a >> 4, (a >> 2) & 3, a & 3
        for j in range(4):  # kq = b_lo
            x = al ^ j
            par = rp(al, j) ^ (pc(al & j & NEG_L) & 1) ^ (pc(u) & (pc(j) + pc(ah)) & 1) ^ (pc(ah) & pc(j) & 1)
            val = 0.0 if (al & j & Z_L) else (-A[a] if par else A[a])
            Aop[u, x, ah, j] = val
    for b in range(64):
        v, bh, bl = b >> 4, (b >> 2) & 3, b & 3
        for s in range(4):  # a_hi
            y = bh ^ s
            par = rp(s, bh) ^ (pc(s & bh & NEG_H) & 1)
            val = 0.0 if (s & bh & Z_H) else (-B[b] if par else B[b])
            Bop[s, bl, v, y] = val
    T = np.einsum("uxsk,skvy->uxvy", Aop, Bop)
    C = np.zeros(64)
    for w, y, x in itertools.product(range(4), repeat=3):
        acc = 0.0
        for u in range(4):
            v = u ^ w
            if u & v & Z_T:
                continue
            par = (pc(u) & pc(y) & 1) ^ rp(u, v) ^ (pc(u & v & NEG_T) & 1)
            acc += -T[u, x, v, y] if par else T[u, x, v, y]
        C[(w << 4) | (y << 2) | x] = acc
    return C


if __name__ == "__main__":
    rng = np.random.default_rng(1)
    worst = 0.0
    for trial in range(40):
        neg = int(rng.integers(0, 64))
        zero = int(rng.integers(0, 64)) & ~neg if trial % 2 else 0
        A, B = rng.uniform(-1, 1, 64), rng.uniform(-1, 1, 64)
        d = np.abs(direct(A, B, neg, zero) - tiled(A, B, neg, zero)).max()
        worst = max(worst, d)
        assert d < 1e-13, (trial, neg, zero, d)
    print("mfma6 tile decomposition agrees with the bitmask product on 40 random (metric, operands) cases; worst |diff| = %.2e" % worst)
