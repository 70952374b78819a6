"""Programs behind tests/golden/golden_eval.json (shared by the generator and the tests)."""
from helpers import full_grades

CGA = [1.0, 1.0, 1.0, 1.0, -1.0]


def _cfg1(B):
    a, b, c = (B.input(s, full_grades(3), 3) for s in range(3))
    return (a + b * c).g(2)


def _sandwich(B):
    r = B.input(0, [0, 2, 4], 5)
    x = B.input(1, [1], 5)
    return r * x * r.rev()


def _sandwich_g1(B):
    return _sandwich(B).g(1)


def _pga_point_join(B):   # PGA2 (metric [0,1,1], degenerate e0): line through two points, its square, a dot
    p = B.input(0, [1], 3)
    q = B.input(1, [1], 3)
    l = p ^ q
    return l * l.rev() + (p & q)


def _r4_mixed(B):
    a = B.input(0, full_grades(4), 4)
    b = B.input(1, [1, 2], 4)
    return ((a << b) * (a >> b)).rev() * (a ^ b).ginvol()


def _sta_norm(B):
    v = B.input(0, [1], 4)
    return v * v.vinv() + v.norm_sq()


def _r5_gp(B):
    return B.input(0, full_grades(5), 5) * B.input(1, full_grades(5), 5)


PROGRAMS = {
    "cfg1_r3": dict(metric=3, dim=3, batch=6, seed=101, build=_cfg1, inputs={0: full_grades(3), 1: full_grades(3), 2: full_grades(3)}),
    "cfg5_sandwich": dict(metric=CGA, dim=5, batch=5, seed=105, build=_sandwich, inputs={0: [0, 2, 4], 1: [1]}),
    "cfg5_sandwich_shared_rotor_g1": dict(metric=CGA, dim=5, batch=5, seed=106, build=_sandwich_g1, inputs={0: [0, 2, 4], 1: [1]}, shared=(0,)),
    "pga2_join": dict(metric=[0.0, 1.0, 1.0], dim=3, batch=4, seed=107, build=_pga_point_join, inputs={0: [1], 1: [1]}),
    "r4_contractions": dict(metric=4, dim=4, batch=4, seed=108, build=_r4_mixed, inputs={0: full_grades(4), 1: [1, 2]}),
    "sta_inverse": dict(metric=[1.0, -1.0, -1.0, -1.0], dim=4, batch=4, seed=109, build=_sta_norm, inputs={0: [1]}),
    "r5_full_gp": dict(metric=5, dim=5, batch=2, seed=110, build=_r5_gp, inputs={0: full_grades(5), 1: full_grades(5)}),
}
