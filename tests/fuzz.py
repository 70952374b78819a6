"""Random programs over the reference's operator surface, for the fuzz tests.

A program is generated once as plain data (`spec`) from a seed and then realised on a back end
(`OracleBackend` = the checker, `HipBackend` = the product), so both see the same expression with
the same leaf values.  Sums are built from two sub-programs of the same shape: the reference panics
on sums of operands whose grade sets differ (quirk Q5), which would otherwise be most of
what a blind generator produces; programs that still make the reference panic are kept -- the product
must then fail too.
"""
from __future__ import annotations

import numpy as np

METRICS = {
    2: [[1.0, 1.0], [1.0, -1.0]],
    3: [[1.0, 1.0, 1.0], [0.0, 1.0, 1.0], [1.0, 1.0, -1.0]],
    4: [[1.0] * 4, [1.0, -1.0, -1.0, -1.0], [0.0, 1.0, 1.0, 1.0], [2.0, -0.5, 3.0, 0.0]],
    5: [[1.0] * 5, [1.0, 1.0, 1.0, 1.0, -1.0]],
    6: [[1.0] * 6, [1.0, -1.0, 1.0, -1.0, 1.0, 1.0]],
}
BIN = ["gp", "gp", "gp", "outer", "inner", "lc", "rc"]
UN = ["neg", "rev", "ginvol", "conj"]


def _leaf(rng, n):
    kind = rng.integers(0, 10)
    if kind == 0:
        return ("scalar", float(np.round(rng.uniform(-2, 2), 3)))
    if kind == 1:
        return ("basis", int(rng.integers(0, n)))
    ng = int(rng.integers(1, min(n + 1, 4) + 1))
    grades = sorted(int(g) for g in rng.choice(n + 1, size=ng, replace=False))
    return ("mv", grades, int(rng.integers(0, 2 ** 31)))


def _gen(rng, n, depth):
    if depth <= 0 or rng.random() < 0.2:
        return _leaf(rng, n)
    r = rng.random()
    if r < 0.45:
        return ("bin", BIN[int(rng.integers(0, len(BIN)))], _gen(rng, n, depth - 1), _gen(rng, n, depth - 1))
    if r < 0.60:
        a = _gen(rng, n, depth - 1)
        return ("sum", "+" if rng.random() < 0.6 else "-", a, _reseed(rng, a))
    if r < 0.78:
        return ("un", UN[int(rng.integers(0, len(UN)))], _gen(rng, n, depth - 1))
    if r < 0.88:
        return ("g", int(rng.integers(0, n + 1)), _gen(rng, n, depth - 1))
    if r < 0.93:
        return ("shared", _gen(rng, n, depth - 1))           # p * p: one node used twice
    if r < 0.97:
        return ("scal", "sinv" if rng.random() < 0.5 else "sqrt", _gen(rng, n, depth - 1))
    return ("divs", float(rng.choice([2.0, -4.0, 0.5])), _gen(rng, n, depth - 1))


def _reseed(rng, spec):
    """same shape, fresh leaf values"""
    if spec[0] == "mv":
        return ("mv", spec[1], int(rng.integers(0, 2 ** 31)))
    if spec[0] in ("scalar", "basis"):
        return spec if spec[0] == "basis" else ("scalar", float(np.round(rng.uniform(-2, 2), 3)))
    return tuple(_reseed(rng, s) if isinstance(s, tuple) else s for s in spec)


def random_big_program(seed):
    """(n, metric, spec) with n = 7, 8 and dense leaves: too big to fuse, so the plan is a sequence of
    launches (exact list kernels, copies, sign flips) even without GAAST_FLAG_NO_FUSION"""
    rng = np.random.default_rng(seed)
    n = int(rng.choice([7, 8]))
    metric = [1.0] * n if rng.random() < 0.5 else [float(rng.choice([1.0, -1.0])) for _ in range(n)]
    if rng.random() < 0.25:
        metric[int(rng.integers(0, n))] = 0.0            # degenerate: general coefficients, CSR kernel
    if rng.random() < 0.35:                               # general diagonal metric (algebra.rs:148-165): squares other than +-1
        for _ in range(int(rng.integers(1, 4))):
            i = int(rng.integers(0, n))
            metric[i] = metric[i] * float(rng.choice([2.0, 0.5, 3.0, 1.5, 0.25])) if metric[i] != 0.0 else 0.0

    def leaf():
        if rng.random() < 0.5:
            grades = list(range(n + 1))
        else:
            grades = sorted(int(g) for g in rng.choice(n + 1, size=int(rng.integers(3, n + 1)), replace=False))
        return ("mv", grades, int(rng.integers(0, 2 ** 31)))

    def gen(depth):
        if depth == 0:
            return leaf()
        r = rng.random()
        if r < 0.55:
            return ("bin", BIN[int(rng.integers(0, len(BIN)))], gen(depth - 1), gen(depth - 1))
        if r < 0.70:
            a = gen(depth - 1)
            return ("sum", "+" if rng.random() < 0.5 else "-", a, _reseed(rng, a))
        if r < 0.90:
            return ("un", UN[int(rng.integers(0, len(UN)))], gen(depth - 1))
        return ("g", int(rng.integers(0, n + 1)), gen(depth - 1))

    return n, metric, ("bin", "gp", gen(int(rng.integers(0, 2))), gen(int(rng.integers(0, 2))))


def random_program(seed):
    """(n, metric, spec)"""
    rng = np.random.default_rng(seed)
    n = int(rng.choice([2, 3, 3, 4, 4, 5, 6]))
    metric = METRICS[n][int(rng.integers(0, len(METRICS[n])))]
    depth = int(rng.integers(2, 5 if n <= 4 else 4))
    return n, metric, _gen(rng, n, depth)


def realise(spec, B, n, rows=None, batch=0):
    """build the expression on back end B.  With `rows` (a dict), multivector leaves become bound inputs
    (B.input) and rows[slot] receives their [batch, row_len] values instead of embedded constants."""
    from helpers import n_choose_k
    kind = spec[0]
    if kind == "scalar":
        return B.scalar(spec[1])
    if kind == "basis":
        return B.basis_vectors(n)[spec[1]]
    if kind == "mv":
        r = np.random.default_rng(spec[2])
        if rows is not None:
            slot = len(rows)
            rows[slot] = r.uniform(-1.0, 1.0, (batch, sum(n_choose_k(n, k) for k in spec[1])))
            return B.input(slot, spec[1], n)
        return B.value({k: r.uniform(-1.0, 1.0, n_choose_k(n, k)) for k in spec[1]}, n)
    if kind == "bin":
        a, b = realise(spec[2], B, n, rows, batch), realise(spec[3], B, n, rows, batch)
        return {"gp": lambda: a * b, "outer": lambda: a ^ b, "inner": lambda: a & b,
                "lc": lambda: a << b, "rc": lambda: a >> b}[spec[1]]()
    if kind == "sum":
        a, b = realise(spec[2], B, n, rows, batch), realise(spec[3], B, n, rows, batch)
        return a + b if spec[1] == "+" else a - b
    if kind == "un":
        a = realise(spec[2], B, n, rows, batch)
        return {"neg": lambda: -a, "rev": a.rev, "ginvol": a.ginvol, "conj": a.conj}[spec[1]]()
    if kind == "g":
        return realise(spec[2], B, n, rows, batch).g(spec[1])
    if kind == "shared":
        p = realise(spec[1], B, n, rows, batch)
        return p * p
    if kind == "scal":
        s = realise(spec[2], B, n, rows, batch).norm_sq()                 # grade 0 only, as sinv / sqrt require
        return s.sinv() if spec[1] == "sinv" else s.sqrt()
    if kind == "divs":
        return realise(spec[2], B, n, rows, batch) / spec[1]
    raise ValueError(kind)
