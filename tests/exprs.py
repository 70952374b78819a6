"""A catalogue of expressions used by both the CPU (host-logic) and GPU (parity) tests.

Each case: name -> (metric, builder(B, rng) -> expr).  Builders draw their fixed values from
`rng` so that the oracle and the product see identical inputs when given equal seeds.
The first four are the reference's own known-answer tests (src/eval.rs:134-163).
"""
from __future__ import annotations

import numpy as np

from helpers import full_grades, random_mv

EGA3 = 3
PGA2 = [0.0, 1.0, 1.0]
CGA = [1.0, 1.0, 1.0, 1.0, -1.0]
STA = [1.0, -1.0, -1.0, -1.0]


def _kat_bivec(B, rng):
    e1, e2, _ = B.basis_vectors(3)
    return e1 ^ e2


def _kat_trivec(B, rng):
    e1, e2, e3 = B.basis_vectors(3)
    return e2 ^ e1 ^ e3


def _kat_norm(B, rng):
    e0, e1, e2 = B.basis_vectors(3)
    return (e0 - 2 * e1 + e2).norm_sq()


def _kat_projection(B, rng):
    e1, e2, e3 = B.basis_vectors(3)
    v = e1 + e2
    bv = 4 * e1 ^ e3
    return (v & bv) & bv.vinv()


def _cfg1(B, rng):  # BASELINE config 1: d = (a + b*c).g(2) on full R^3 multivectors
    a, b, c = (B.value(random_mv(rng, 3, full_grades(3)), 3) for _ in range(3))
    return (a + b * c).g(2)


def _sandwich(B, rng):  # BASELINE config 5: R X ~R in R^{4,1}
    r = B.value(random_mv(rng, 5, [0, 2, 4]), 5)
    x = B.value(random_mv(rng, 5, [1]), 5)
    return r * x * r.rev()


def _sandwich_g1(B, rng):
    return _sandwich(B, rng).g(1)


def _all_products(kind):
    def build(B, rng):
        a = B.value(random_mv(rng, 4, full_grades(4)), 4)
        b = B.value(random_mv(rng, 4, full_grades(4)), 4)
        return {"gp": lambda: a * b, "outer": lambda: a ^ b, "inner": lambda: a & b,
                "lc": lambda: a << b, "rc": lambda: a >> b}[kind]()
    return build


def _unary_chain(B, rng):
    a = B.value(random_mv(rng, 4, full_grades(4)), 4)
    b = B.value(random_mv(rng, 4, [1, 2]), 4)
    return (-a).rev() * b.ginvol() + a.conj()


def _q1_sub(B, rng):  # e1 - e2 evaluates to -(e1 + e2) in the reference
    e1, e2, _ = B.basis_vectors(3)
    return e1 - e2


def _shared_subexpr(B, rng):
    a = B.value(random_mv(rng, 3, [1]), 3)
    b = B.value(random_mv(rng, 3, [1]), 3)
    p = a * b
    return p * p + p.g(0)


def _scalar_ops(B, rng):
    a = B.value(random_mv(rng, 3, [1], 0.5, 2.0), 3)
    return a.norm_sq().sqrt() + a.norm_sq().sinv()


def _vinv_vector(B, rng):
    a = B.value(random_mv(rng, 4, [1], 0.5, 2.0), 4)
    return a * a.vinv()


def _div_scalar(B, rng):
    a = B.value(random_mv(rng, 3, full_grades(3)), 3)
    return (a * a) / 4


def _zero_literal(B, rng):
    a = B.value(random_mv(rng, 3, [1, 2]), 3)
    return a + 0 * a + 0


def _q3_overapprox(B, rng):  # e123*e123 keeps an all-zero grade-2 slab
    t = B.value({3: [2.0]}, 3)
    return t * t


def _sta_rotor(B, rng):
    a = B.value(random_mv(rng, 4, [1]), 4)
    b = B.value(random_mv(rng, 4, [1]), 4)
    r = a * b
    x = B.value(random_mv(rng, 4, [1]), 4)
    return (r * x * r.rev()).g(1)


def _gp_full(n):
    def build(B, rng):
        a = B.value(random_mv(rng, n, full_grades(n)), n)
        b = B.value(random_mv(rng, n, full_grades(n)), n)
        return a * b
    return build


def _gp_even(n):
    def build(B, rng):
        ev = [k for k in range(n + 1) if k % 2 == 0]
        a = B.value(random_mv(rng, n, ev), n)
        b = B.value(random_mv(rng, n, ev), n)
        return a * b
    return build


def _pga_motor(B, rng):
    m = B.value(random_mv(rng, 3, [0, 2]), 3)
    p = B.value(random_mv(rng, 3, [1]), 3)
    return m * p * m.rev()


def _gselect(B, rng):
    a = B.value(random_mv(rng, 5, full_grades(5)), 5)
    b = B.value(random_mv(rng, 5, full_grades(5)), 5)
    return (a * b).gselect([0, 3, 5])


CASES = {
    "kat_vecs_to_bivec": (EGA3, _kat_bivec),
    "kat_vecs_to_trivec": (EGA3, _kat_trivec),
    "kat_vec_norm": (PGA2, _kat_norm),
    "kat_projection": (EGA3, _kat_projection),
    "cfg1_r3": (EGA3, _cfg1),
    "cfg5_sandwich": (CGA, _sandwich),
    "cfg5_sandwich_g1": (CGA, _sandwich_g1),
    "r4_gp": (4, _all_products("gp")),
    "r4_outer": (4, _all_products("outer")),
    "r4_inner": (4, _all_products("inner")),
    "r4_lcontract": (4, _all_products("lc")),
    "r4_rcontract": (4, _all_products("rc")),
    "sta_gp": (STA, _all_products("gp")),
    "unary_chain": (4, _unary_chain),
    "q1_sub": (EGA3, _q1_sub),
    "shared_subexpr": (EGA3, _shared_subexpr),
    "scalar_ops": (EGA3, _scalar_ops),
    "vinv_vector": ([1.0, 1.0, -1.0, 1.0], _vinv_vector),
    "div_scalar": (EGA3, _div_scalar),
    "zero_literal": (EGA3, _zero_literal),
    "q3_overapprox": (EGA3, _q3_overapprox),
    "sta_rotor": (STA, _sta_rotor),
    "pga_motor": (PGA2, _pga_motor),
    "gselect": (5, _gselect),
    "r5_gp_full": (5, _gp_full(5)),
    "r6_gp_full": (6, _gp_full(6)),
    "r6_gp_even": (6, _gp_even(6)),
    "weird_metric_gp": ([2.0, -0.5, 3.0, 0.0], _all_products("gp")),
}
